#!/bin/bash
# The exact command lists of round 5's GPU sessions (each ran on an MI355X box through gpurun; outputs under gpurun_out/, the summaries that are
# evidence were copied to profiles/r05*).  usage: tools/gpu_r05_calls.sh <n>      e.g.  gpurun --timeout 1200 -- 'bash tools/gpu_r05_calls.sh 1'
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD

# round 5, GPU call 1: (a) may a VALU op overwrite the A / B operand of the MFMA issued just before it (the x3 GEMM's 8-product variant has that
# pattern)?  (b) x3 GEMM design space: ablations of the product structure, ring of three, 64-deep K per barrier, 256 x 128 block, W through LDS;
# (c) the RCCL path at world size 1 (test + one bench step through the collectives); (d) WHAT differs in the not-bit-stable forms (stage images).
call1() {
    O=$PWD/gpurun_out/r05a
    mkdir -p $O
    timeout 200 tools/microbench/bin/mfma_war_hazard > $O/mfma_war_hazard.log 2>&1; echo "war_hazard rc=$?" | tee $O/status.txt
    grep -v " 0 of " $O/mfma_war_hazard.log | head -20; echo "(lines with 0 differences: $(grep -c ' 0 of ' $O/mfma_war_hazard.log))"
    timeout 400 tools/microbench/bin/x3_gemm_lab > $O/x3_gemm_lab.log 2>&1; echo "x3_gemm_lab rc=$?" | tee -a $O/status.txt
    cat $O/x3_gemm_lab.log
    timeout 300 python -m pytest tests/test_gpu_dist.py -x -q > $O/pytest_dist.log 2>&1; echo "pytest dist rc=$?" | tee -a $O/status.txt; tail -3 $O/pytest_dist.log
    ITTS_BENCH_FORCE_DIST=1 timeout 400 python bench.py --steps 1 --warmup 1 --alt-steps 0 --no-configs --no-shards --no-cpu-baseline --no-extras > $O/bench_force_dist.json 2> $O/bench_force_dist.err; echo "bench force_dist rc=$?" | tee -a $O/status.txt
    python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r05a/bench_force_dist.json").read().strip().splitlines()[-1])
    print("bench (FORCE_DIST):", j["value"], j["unit"], "ms/step", j["ms_per_step"])
except Exception as e:
    print("bench parse failed", e)
PY
    tail -3 $O/bench_force_dist.err
    timeout 500 python tools/s2mel_capture.py 2 517 1926 12 bf16:s2mel_fused=2 fp32x3:x3_attn=0,x3_products=8,x3_pin=0 > $O/capture.log 2>&1; echo "capture rc=$?" | tee -a $O/status.txt
    head -120 $O/capture.log
}

# round 5, GPU call 2: (a) the epilogue's SLP-packed RoPE sequence verbatim beside a busy neighbour block (in-place v_pk_mul_f32 vs a separate
# destination); (b) the formerly unstable forms with the RoPE on scalar fmas: 24 stage traces each; (c) the causal MFMA prefill attention: GPT tests,
# prefill time at 64 rows; (d) x3 GEMM lab, second round (scalar addressing, tile-group sizes, W-through-LDS ablations); (e) s2mel / x3 GEMM tests.
call2() {
    O=$PWD/gpurun_out/r05b
    mkdir -p $O
    timeout 300 tools/microbench/bin/pk_inplace_hazard > $O/pk_inplace_hazard.log 2>&1; echo "pk_inplace rc=$?" | tee $O/status.txt
    cat $O/pk_inplace_hazard.log
    timeout 900 python tools/s2mel_trace.py 2 517 1926 24 bf16:s2mel_fused=2 fp32x3:x3_attn=0,x3_products=8,x3_pin=0 fp32x3:x3_attn=0,x3_products=8,x3_sched=0,x3_pin=0 fp32x3:x3_attn=0,x3_sched=0,x3_pin=0 fp32x3:x3_products=8,x3_pin=0 fp32x3 > $O/trace.log 2>&1; echo "trace rc=$?" | tee -a $O/status.txt
    grep -E "repetitions agree|histogram" $O/trace.log
    timeout 400 tools/microbench/bin/x3_gemm_lab > $O/x3_gemm_lab2.log 2>&1; echo "x3_gemm_lab rc=$?" | tee -a $O/status.txt
    cat $O/x3_gemm_lab2.log
    timeout 300 python tools/decode_bench.py 40 1,64 prefill_attn=0 > $O/prefill_bench.log 2>&1; echo "prefill bench rc=$?" | tee -a $O/status.txt
    grep "^B=" $O/prefill_bench.log
    timeout 1200 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_edges.py tests/test_gpu_compaction.py -x -q -s > $O/pytest_gpt.log 2>&1; echo "pytest gpt rc=$?" | tee -a $O/status.txt
    grep -E "prefill MFMA|passed|failed|Error|error" $O/pytest_gpt.log | tail -8
    timeout 1200 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_gemm_x3.py -x -q > $O/pytest_s2mel.log 2>&1; echo "pytest s2mel rc=$?" | tee -a $O/status.txt
    tail -4 $O/pytest_s2mel.log
}

# round 5, GPU call 3: the product x3 GEMM with scalar addressing (pf_uni): GEMM shapes, solve time (with / without plane operands), tests of
# everything the RoPE / addressing / fusion changes touch (GPT incl. the prefill attention, s2mel incl. determinism of every variant, x3 GEMM).
call3() {
    O=$PWD/gpurun_out/r05c
    mkdir -p $O
    timeout 300 python tools/gemm_x3_bench.py 312704 5 > $O/gemm_x3_bench.log 2>&1; echo "gemm_x3_bench rc=$?" | tee $O/status.txt
    cat $O/gemm_x3_bench.log
    timeout 600 python tools/s2mel_bench.py 8 517 1926 3 fp32x3 fp32x3:x3_aplanes=1 fp32x3 fp32x3:x3_aplanes=1 bf16 fp32 > $O/s2mel_bench.log 2>&1; echo "s2mel_bench rc=$?" | tee -a $O/status.txt
    grep "^B=" $O/s2mel_bench.log
    timeout 1200 python -m pytest tests/test_gpu_gpt.py -x -q -s -k "prefill or latent or bf16 or golden" > $O/pytest_gpt.log 2>&1; echo "pytest gpt rc=$?" | tee -a $O/status.txt
    grep -E "prefill MFMA|passed|failed|Error|error" $O/pytest_gpt.log | tail -8
    timeout 1500 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_gemm_x3.py -x -q > $O/pytest_s2mel.log 2>&1; echo "pytest s2mel rc=$?" | tee -a $O/status.txt
    tail -4 $O/pytest_s2mel.log
}

# round 5, GPU call 4: (a) which arithmetic of the tiny spk_emb_proj projection keeps the reference-minted `typical_greedy` fixture's borderline
# token (engine f32 GEMM / f64 on the host / vendor BLAS); (b) the bf16 x 3 vocoder conv: unit op vs f64 and the f32 kernel, generator vs the
# reference-class waveforms, vocoder forward at 16 x 1926 frames in the three conv modes with per-stage conv rates.
call4() {
    O=$PWD/gpurun_out/r05d
    mkdir -p $O
    for m in engine f64 blas; do
        ITTS_SPK_PROJ=$m timeout 600 python -m pytest tests/test_gpu_gpt.py -q -k "codes_bit_exact or beam_codes_bit_exact or v1_decode or v2_decode" > $O/pytest_spk_$m.log 2>&1; echo "spk proj $m rc=$? $(tail -1 $O/pytest_spk_$m.log)" | tee -a $O/status.txt
    done
    timeout 900 python -m pytest tests/test_gpu_bigvgan_x3.py -x -q -s > $O/pytest_voc_x3.log 2>&1; echo "pytest voc x3 rc=$?" | tee -a $O/status.txt
    grep -E "x3 conv|rms err|passed|failed|Error|error" $O/pytest_voc_x3.log | tail -20
    timeout 600 python tools/voc_h3_bench.py 16 f32,bf16x3:96,f16x3:96 > $O/voc_bench.log 2>&1; echo "voc bench rc=$?" | tee -a $O/status.txt
    cat $O/voc_bench.log
}

# round 5, GPU call 5: the fp32x3 GEMM from its own translation unit (no SLP, plain subtractions): GEMM shapes, solve time, s2mel + x3 GEMM tests;
# GPT fixtures with the f64 speaker projection; vocoder x3 tests again (ragged fixture fixed).
call5() {
    O=$PWD/gpurun_out/r05e
    mkdir -p $O
    timeout 300 python tools/gemm_x3_bench.py 312704 5 > $O/gemm_x3_bench.log 2>&1; echo "gemm_x3_bench rc=$?" | tee $O/status.txt
    grep f32x3 $O/gemm_x3_bench.log
    timeout 600 python tools/s2mel_bench.py 8 517 1926 3 fp32x3 fp32x3:x3_aplanes=1 fp32x3 fp32x3:x3_aplanes=1 > $O/s2mel_bench.log 2>&1; echo "s2mel_bench rc=$?" | tee -a $O/status.txt
    grep "^B=" $O/s2mel_bench.log
    timeout 1500 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_gemm_x3.py tests/test_gpu_attn_x3.py -x -q > $O/pytest_s2mel.log 2>&1; echo "pytest s2mel rc=$?" | tee -a $O/status.txt
    tail -4 $O/pytest_s2mel.log
    timeout 1500 python -m pytest tests/test_gpu_gpt.py -x -q > $O/pytest_gpt.log 2>&1; echo "pytest gpt rc=$?" | tee -a $O/status.txt
    tail -4 $O/pytest_gpt.log
    timeout 900 python -m pytest tests/test_gpu_bigvgan_x3.py tests/test_gpu_bigvgan.py tests/test_gpu_bigvgan_h3.py -x -q > $O/pytest_voc.log 2>&1; echo "pytest voc rc=$?" | tee -a $O/status.txt
    tail -4 $O/pytest_voc.log
}

# round 5, GPU call 6: admission into a running decode batch; the two-co-tile form of the x3 vocoder conv (tests + forward time); GPT fixtures with the
# sampler's row-relative step; then the benchmark as the driver runs it, and the rocprofv3 kernel stats of a short run.
call6() {
    O=$PWD/gpurun_out/r05f
    mkdir -p $O
    timeout 600 python -m pytest tests/test_gpu_admission.py -x -q -s > $O/pytest_admission.log 2>&1; echo "pytest admission rc=$?" | tee $O/status.txt
    grep -E "admitted at|passed|failed|Error|error" $O/pytest_admission.log | tail -8
    timeout 900 python -m pytest tests/test_gpu_bigvgan_x3.py tests/test_gpu_bigvgan.py -x -q -s > $O/pytest_voc.log 2>&1; echo "pytest voc rc=$?" | tee -a $O/status.txt
    grep -E "x3 conv|rms err|passed|failed|Error|error" $O/pytest_voc.log | tail -16
    timeout 600 python tools/voc_h3_bench.py 16 f32,bf16x3:96 > $O/voc_bench.log 2>&1; echo "voc bench rc=$?" | tee -a $O/status.txt
    cat $O/voc_bench.log
    timeout 1500 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_compaction.py tests/test_gpu_edges.py tests/test_streaming.py -x -q > $O/pytest_gpt.log 2>&1; echo "pytest gpt rc=$?" | tee -a $O/status.txt
    tail -3 $O/pytest_gpt.log
    timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt
    python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r05f/bench.json").read().strip().splitlines()[-1])
    st = j["stages"]
    print("bench:", round(j["value"], 2), j["unit"], "ms/step", round(j["ms_per_step"], 1), "| roofline frac", round(j["roofline"]["frac"], 3), j["roofline"]["achieved"])
    print("  gpt prefill", round(st["gpt_prefill_ms_per_step"], 1), "decode", round(st["gpt_decode_ms_per_step"], 1), "ms/token", round(st["gpt_decode_ms_per_token"], 3), "bigvgan", round(st["bigvgan_ms_per_step"], 1))
    s2 = st["s2mel"]
    print("  s2mel:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in s2.items() if k.startswith("cfm_") or k.endswith("ms_per_step")})
except Exception as e:
    print("bench parse failed", e)
PY
    tail -3 $O/bench.err
}

# round 5, GPU call 7: admission test (fixed inputs); SAME-BOX A/B of round 4's library (tools/ab/r04_tree, built from commit 43a1160, not committed)
# against this round's: the 64-utterance flow-matching solve and the vocoder forward; the benchmark with the bf16 x 3 vocoder convs; rocprofv3
# kernel stats of a one-step run; the matrix-pipe PMC pass with per-XCD-normalised fields.
call7() {
    O=$PWD/gpurun_out/r05g
    mkdir -p $O
    timeout 600 python -m pytest tests/test_gpu_admission.py -x -q -s > $O/pytest_admission.log 2>&1; echo "pytest admission rc=$?" | tee $O/status.txt
    grep -E "admitted at|passed|failed|Error|error" $O/pytest_admission.log | tail -8
    for rep in 1 2; do
        (cd tools/ab/r04_tree && timeout 300 python tools/s2mel_bench.py 64 517 1926 1 fp32x3) > $O/ab_r04_solve_$rep.log 2>&1; grep "^B=" $O/ab_r04_solve_$rep.log | sed 's/^/r04: /'
        timeout 300 python tools/s2mel_bench.py 64 517 1926 1 fp32x3 > $O/ab_r05_solve_$rep.log 2>&1; grep "^B=" $O/ab_r05_solve_$rep.log | sed 's/^/r05: /'
    done
    (cd tools/ab/r04_tree && timeout 300 python tools/voc_h3_bench.py 64 f32) > $O/ab_r04_voc.log 2>&1; grep "^B=" $O/ab_r04_voc.log | sed 's/^/r04: /'
    timeout 300 python tools/voc_h3_bench.py 64 f32,bf16x3 > $O/ab_r05_voc.log 2>&1; grep "^B=" $O/ab_r05_voc.log | sed 's/^/r05: /'
    echo "ab rc=0" | tee -a $O/status.txt
    timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.txt
    python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r05g/bench.json").read().strip().splitlines()[-1])
    st = j["stages"]
    print("bench:", round(j["value"], 2), j["unit"], "ms/step", round(j["ms_per_step"], 1), "| roofline frac", round(j["roofline"]["frac"], 3), round(j["roofline"]["achieved"], 1))
    print("  gpt prefill", round(st["gpt_prefill_ms_per_step"], 1), "decode", round(st["gpt_decode_ms_per_step"], 1), "ms/token", round(st["gpt_decode_ms_per_token"], 3), "bigvgan", round(st["bigvgan_ms_per_step"], 1))
    s2 = st["s2mel"]
    print("  s2mel:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in s2.items() if k.startswith("cfm_") or k.endswith("ms_per_step")})
    print("  other conv mode:", st.get("bigvgan_other_conv_mode"))
except Exception as e:
    print("bench parse failed", e)
PY
    cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --alt-steps 0 --no-configs --no-shards --no-cpu-baseline --no-extras > $O/bench_profiled_run.json 2> $O/bench_profiled.err; echo "rocprof rc=$?" | tee -a $O/status.txt
    cd $ROOT
    cp "$(find $O/prof -name '*kernel_stats.csv' | head -1)" $O/bench_kernel_stats.csv 2>/dev/null; rm -rf $O/prof
    head -14 $O/bench_kernel_stats.csv | cut -c1-150
    timeout 900 bash tools/pmc_x3.sh 8 > $O/pmc_x3.log 2>&1; echo "pmc_x3 rc=$?" | tee -a $O/status.txt; cp gpurun_out/pmc_x3/x3_pmc.json $O/ 2>/dev/null
    python -c "
import json; j=json.load(open('$O/x3_pmc.json'))
for k,v in j.items():
    if isinstance(v, dict): print(k, 'busy', round(v['mfma_busy_of_own_cycles'],3), 'clock', round(v['effective_clock_GHz'],3))
" 2>/dev/null
}

# round 5, GPU call 8: admission test; rocprofv3 kernel stats of a one-step bench run (csv)
call8() {
    O=$PWD/gpurun_out/r05h
    mkdir -p $O
    timeout 600 python -m pytest tests/test_gpu_admission.py -x -q -s > $O/pytest_admission.log 2>&1; echo "pytest admission rc=$?" | tee $O/status.txt
    grep -E "admitted at|passed|failed|Error|error" $O/pytest_admission.log | tail -8
    timeout 1200 bash tools/profile.sh r05h --alt-steps 0 --no-configs --no-shards --no-extras > $O/profile.log 2>&1; echo "profile rc=$?" | tee -a $O/status.txt
    cp gpurun_out/prof_r05h/kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null; cp gpurun_out/prof_r05h/bench.json $O/bench_profiled_run.json 2>/dev/null
    head -16 $O/bench_kernel_stats.csv | cut -c1-170
}

# round 5, GPU call 9: the whole GPU suite + smoke at the current code
call9() {
    O=$PWD/gpurun_out/r05i
    mkdir -p $O
    timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee $O/status.txt
    tail -6 $O/pytest_gpu.log
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
    tail -4 $O/smoke.log
}

# round 5, GPU call 10: x3 vocoder conv with the next tile's window fragments read under the MFMAs: tests + forward time
call10() {
    O=$PWD/gpurun_out/r05j
    mkdir -p $O
    timeout 900 python -m pytest tests/test_gpu_bigvgan_x3.py -x -q -s > $O/pytest_voc.log 2>&1; echo "pytest voc rc=$?" | tee $O/status.txt
    grep -E "x3 conv|rms err|passed|failed|Error|error" $O/pytest_voc.log | tail -14
    timeout 600 python tools/voc_h3_bench.py 16 bf16x3:96,f32 > $O/voc_bench.log 2>&1; echo "voc bench rc=$?" | tee -a $O/status.txt
    cat $O/voc_bench.log
}

# round 5, GPU call 11: operand split on v_dot2c_f32_bf16 (7 instead of 11 VALU ops per pair): bit check of the planes, x3 tests, same-box A/B
# against the previous build (tools/ab/base_tree, not committed)
call11() {
    O=$PWD/gpurun_out/r05k
    mkdir -p $O
    timeout 300 tools/microbench/bin/split_dot2 > $O/split_dot2.log 2>&1; echo "split_dot2 rc=$?" | tee $O/status.txt
    cat $O/split_dot2.log
    timeout 900 python -m pytest tests/test_gpu_gemm_x3.py tests/test_gpu_attn_x3.py tests/test_gpu_bigvgan_x3.py -x -q > $O/pytest_x3.log 2>&1; echo "pytest x3 rc=$?" | tee -a $O/status.txt
    tail -3 $O/pytest_x3.log
    timeout 900 python -m pytest tests/test_gpu_s2mel.py -x -q > $O/pytest_s2mel.log 2>&1; echo "pytest s2mel rc=$?" | tee -a $O/status.txt
    tail -3 $O/pytest_s2mel.log
    for rep in 1 2; do
        timeout 300 python tools/gemm_x3_bench.py > $O/gemm_x3_bench_$rep.log 2>&1; grep "^M=" $O/gemm_x3_bench_$rep.log | sed 's/^/dot2: /'
        (cd tools/ab/base_tree && timeout 300 python tools/gemm_x3_bench.py) > $O/ab_base_gemm_$rep.log 2>&1; grep "^M=" $O/ab_base_gemm_$rep.log | sed 's/^/base: /'
        timeout 300 python tools/s2mel_bench.py 64 517 1926 1 fp32x3 > $O/solve_$rep.log 2>&1; grep "^B=" $O/solve_$rep.log | sed 's/^/dot2: /'
        (cd tools/ab/base_tree && timeout 300 python tools/s2mel_bench.py 64 517 1926 1 fp32x3) > $O/ab_base_solve_$rep.log 2>&1; grep "^B=" $O/ab_base_solve_$rep.log | sed 's/^/base: /'
    done
}

# round 5, GPU call 12: in-flight schedule against the engine (inference_speech_inflight == one batch), admission test again
call12() {
    O=$PWD/gpurun_out/r05l
    mkdir -p $O
    timeout 900 python -m pytest tests/test_gpu_admission.py -q -s > $O/pytest_admission.log 2>&1; echo "pytest admission rc=$?" | tee $O/status.txt
    grep -E "in-flight|admitted at|passed|failed|Error|assert" $O/pytest_admission.log | tail -20
}

# round 5, GPU call 13: activation written as the x3 conv's operand planes: bit-identity test, vocoder tests, forward time with / without
call13() {
    O=$PWD/gpurun_out/r05m
    mkdir -p $O
    timeout 900 python -m pytest tests/test_gpu_bigvgan_x3.py -x -q > $O/pytest_voc_x3.log 2>&1; echo "pytest voc x3 rc=$?" | tee $O/status.txt
    tail -5 $O/pytest_voc_x3.log
    timeout 600 python tools/voc_h3_bench.py 16 bf16x3:96,bf16x3:96:voc_act_planes=0 > $O/voc_bench.log 2>&1; echo "voc bench rc=$?" | tee -a $O/status.txt
    cat $O/voc_bench.log
}

# round 5, GPU call 14: per-utterance caps in the decode session, admission tests again, in-flight vs drained batches at production width
call14() {
    O=$PWD/gpurun_out/r05n
    mkdir -p $O
    timeout 900 python -m pytest tests/test_gpu_admission.py -q -s > $O/pytest_admission.log 2>&1; echo "pytest admission rc=$?" | tee $O/status.txt
    grep -E "in-flight|passed|failed|Error" $O/pytest_admission.log | tail -8
    timeout 600 python tools/inflight_bench.py 512 64 32 8 > $O/inflight_bench.log 2>&1; echo "inflight bench rc=$?" | tee -a $O/status.txt
    timeout 600 python tools/inflight_bench.py 512 64 32 8 280 560 >> $O/inflight_bench.log 2>&1
    grep -E "^N=|Error|error" $O/inflight_bench.log | tail -8
}

# round 5, final validation A: the whole GPU suite + smoke at the final code
call15() {
    O=$PWD/gpurun_out/r05o
    mkdir -p $O
    timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee $O/status.txt
    tail -6 $O/pytest_gpu.log
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
    tail -4 $O/smoke.log
}

# round 5, final validation B: the bench line (every leg; 5 timed steps) + rocprofv3 kernel stats of the same command without the extra legs
call16() {
    O=$PWD/gpurun_out/r05p
    mkdir -p $O
    timeout 1500 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee $O/status.txt
    tail -c 1800 $O/bench.json
    timeout 900 bash tools/profile.sh r05p --alt-steps 0 --no-configs --no-shards --no-extras > $O/profile.log 2>&1; echo "profile rc=$?" | tee -a $O/status.txt
    cp gpurun_out/prof_r05p/kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null; cp gpurun_out/prof_r05p/bench.json $O/bench_profiled_run.json 2>/dev/null
    head -12 $O/bench_kernel_stats.csv | cut -c1-150
}

# round 5, GPU call 18: x3 GEMM with the split really interleaved under the MFMAs (per-m-tile scheduling regions, 1 MFMA : 2 VALU): same-box A/B
# against the previous build (tools/ab/base_tree), bits of the solve, x3 GEMM tests, s2mel tests
call18() {
    O=$PWD/gpurun_out/r05r
    mkdir -p $O
    timeout 300 python tools/gemm_x3_bench.py > $O/gemm_x3_bench.log 2>&1; grep "^M=.*f32x3" $O/gemm_x3_bench.log | sed 's/^/new:  /'
    (cd tools/ab/base_tree && timeout 300 python tools/gemm_x3_bench.py) > $O/ab_base_gemm.log 2>&1; grep "^M=.*f32x3" $O/ab_base_gemm.log | sed 's/^/base: /'
    timeout 300 python tools/s2mel_bench.py 64 517 1926 1 fp32x3 > $O/solve.log 2>&1; grep "^B=" $O/solve.log | sed 's/^/new:  /'
    (cd tools/ab/base_tree && timeout 300 python tools/s2mel_bench.py 64 517 1926 1 fp32x3) > $O/ab_base_solve.log 2>&1; grep "^B=" $O/ab_base_solve.log | sed 's/^/base: /'
    timeout 600 python -m pytest tests/test_gpu_gemm_x3.py tests/test_gpu_s2mel.py -x -q > $O/pytest_x3_s2mel.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt
    tail -3 $O/pytest_x3_s2mel.log
}

# round 5, GPU call 19: the exact interleave on the plain GEMM instantiations only (tap-mode conv keeps its form): same-box A/B, bits of the solve
call19() {
    O=$PWD/gpurun_out/r05s
    mkdir -p $O
    timeout 300 python tools/gemm_x3_bench.py > $O/gemm_x3_bench.log 2>&1; grep "^M=.*f32x3" $O/gemm_x3_bench.log | sed 's/^/new:  /'
    (cd tools/ab/base_tree && timeout 300 python tools/gemm_x3_bench.py) > $O/ab_base_gemm.log 2>&1; grep "^M=.*f32x3" $O/ab_base_gemm.log | sed 's/^/base: /'
    for rep in 1 2; do
        timeout 300 python tools/s2mel_bench.py 64 517 1926 1 fp32x3 > $O/solve_$rep.log 2>&1; grep "^B=" $O/solve_$rep.log | sed 's/^/new:  /'
        (cd tools/ab/base_tree && timeout 300 python tools/s2mel_bench.py 64 517 1926 1 fp32x3) > $O/ab_base_solve_$rep.log 2>&1; grep "^B=" $O/ab_base_solve_$rep.log | sed 's/^/base: /'
    done
}

"call$1"
