#!/bin/bash
# The exact command lists of round 6's GPU sessions (each ran on an MI355X box through gpurun; outputs under gpurun_out/, the summaries that are
# evidence were copied to profiles/r06*).  usage: tools/gpu_r06_calls.sh <n>      e.g.  gpurun --timeout 1200 -- 'bash tools/gpu_r06_calls.sh 1'
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD

# round 6, GPU call 1: per-slot cache positions / steps in the decode session (admission at any step, one session per call): the admission tests,
# every GPT test (the sampler, the QKV epilogues and the KV-cache attention changed), the in-flight schedule against drained batches.
call1() {
    O=$PWD/gpurun_out/r06a
    mkdir -p $O
    timeout 900 python -m pytest tests/test_gpu_admission.py tests/test_gpu_compaction.py -x -q -s > $O/pytest_admission.log 2>&1; echo "pytest admission rc=$?" | tee $O/status.txt
    grep -E "admitted at|in-flight schedule|passed|failed|Error|error" $O/pytest_admission.log | tail -12
    timeout 1500 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py -x -q > $O/pytest_gpt.log 2>&1; echo "pytest gpt rc=$?" | tee -a $O/status.txt
    tail -4 $O/pytest_gpt.log
    timeout 600 python tools/inflight_bench.py 512 64 32 8 120 560 > $O/inflight_512_120.log 2>&1; echo "inflight rc=$?" | tee -a $O/status.txt; tail -1 $O/inflight_512_120.log
    timeout 600 python tools/inflight_bench.py 512 64 32 8 280 560 > $O/inflight_512_280.log 2>&1; tail -1 $O/inflight_512_280.log
    timeout 600 python tools/inflight_bench.py 128 64 32 8 120 560 > $O/inflight_128_120.log 2>&1; tail -1 $O/inflight_128_120.log
    timeout 600 python tools/inflight_bench.py 512 64 16 4 120 560 > $O/inflight_512_120_c16.log 2>&1; tail -1 $O/inflight_512_120_c16.log
}

# round 6, GPU call 2: (a) the scheduler on the engine's own flag check (itts_gpt_set_chunk_return): admission tests again, in-flight vs drained;
# (b) x3 GEMM with a raised wave priority on the MFMA / split section (option x3_prio): GEMM shapes and the 64-utterance solve, same box, alternating.
call2() {
    O=$PWD/gpurun_out/r06b
    mkdir -p $O
    timeout 600 python -m pytest tests/test_gpu_admission.py tests/test_gpu_compaction.py -x -q -s > $O/pytest_admission.log 2>&1; echo "pytest admission rc=$?" | tee $O/status.txt
    grep -E "admitted at|in-flight schedule|passed|failed|Error|error" $O/pytest_admission.log | tail -8
    for spec in "512 64 64 8 120 560" "512 64 64 8 280 560" "256 64 64 8 120 560" "512 64 64 4 120 560" "512 64 64 16 120 560"; do
        timeout 600 python tools/inflight_bench.py $spec > $O/inflight_$(echo $spec | tr ' ' '_').log 2>&1; tail -1 $O/inflight_$(echo $spec | tr ' ' '_').log
    done
    for p in 0 1 2 3 0 1; do
        timeout 300 python tools/gemm_x3_bench.py 312704 5 x3_prio=$p > $O/gemm_prio$p.log 2>&1; echo "x3_prio=$p: $(grep 'f32x3' $O/gemm_prio$p.log | sed 's/.*f32x3: //' | awk '{printf "%s ", $3}')"
    done
    timeout 900 python tools/s2mel_bench.py 64 517 1926 1 fp32x3 fp32x3:x3_prio=1 fp32x3:x3_prio=2 fp32x3:x3_prio=3 fp32x3 fp32x3:x3_prio=1 > $O/solve_prio.log 2>&1; echo "solve rc=$?" | tee -a $O/status.txt
    grep "^B=" $O/solve_prio.log
}

# round 6, GPU call 3: the scheduler's single-synchronisation poll; the in-flight schedule at production widths against one batch; in-flight vs drained
call3() {
    O=$PWD/gpurun_out/r06c
    mkdir -p $O
    timeout 900 python -m pytest tests/test_gpu_admission.py -x -q -s > $O/pytest_admission.log 2>&1; echo "pytest admission rc=$?" | tee $O/status.txt
    grep -E "production widths|in-flight schedule|passed|failed|Error|error" $O/pytest_admission.log | tail -8
    for spec in "512 64 64 8 120 560" "512 64 64 8 280 560" "256 64 64 8 120 560"; do
        timeout 600 python tools/inflight_bench.py $spec > $O/inflight_$(echo $spec | tr ' ' '_').log 2>&1; tail -1 $O/inflight_$(echo $spec | tr ' ' '_').log
    done
}

# round 6, validation A (intermediate and final): every GPU test + smoke
call4() {
    O=$PWD/gpurun_out/${TAG:-r06d}
    mkdir -p $O
    timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?" | tee $O/status.txt
    tail -6 $O/pytest_gpu.log
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
    tail -4 $O/smoke.log
}

# round 6, validation B: the bench line (every leg; 5 timed steps) + rocprofv3 kernel stats of the same command without the extra legs
call5() {
    T=${TAG:-r06e}
    O=$PWD/gpurun_out/$T
    mkdir -p $O
    timeout 1500 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee $O/status.txt
    tail -c 1800 $O/bench.json
    timeout 900 bash tools/profile.sh $T --alt-steps 0 --no-configs --no-shards --no-extras > $O/profile.log 2>&1; echo "profile rc=$?" | tee -a $O/status.txt
    cp gpurun_out/prof_$T/kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null; cp gpurun_out/prof_$T/bench.json $O/bench_profiled_run.json 2>/dev/null
    head -12 $O/bench_kernel_stats.csv | cut -c1-150
}

# round 6, GPU call 6: (a) where the timed step's wall time goes that is not kernel time (kernel-trace gap report); (b) decode ms per token at 1 / 8 /
# 64 rows with the position shifts read only after an admission; admission tests
call6() {
    O=$PWD/gpurun_out/r06f
    mkdir -p $O
    timeout 900 bash tools/gap_report.sh r06f > $O/gap_report.log 2>&1; echo "gap report rc=$?" | tee $O/status.txt
    cat gpurun_out/gap_r06f/report.txt | cut -c1-220
    timeout 600 python tools/decode_bench.py 400 1,8,64 > $O/decode_bench.log 2>&1; echo "decode bench rc=$?" | tee -a $O/status.txt; grep "^B=" $O/decode_bench.log
    timeout 600 python -m pytest tests/test_gpu_admission.py tests/test_gpu_compaction.py -x -q > $O/pytest_admission.log 2>&1; echo "pytest admission rc=$?" | tee -a $O/status.txt; tail -2 $O/pytest_admission.log
}

# round 6, GPU call 7: the flow-matching / codec host side with row tables built on the device and vectorised packing: s2mel + codec + pipeline
# tests, then the gap report again
call7() {
    O=$PWD/gpurun_out/r06g
    mkdir -p $O
    timeout 1500 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_codec.py tests/test_gpu_cond.py tests/test_gpu_pipeline.py tests/test_gpu_pipeline_fullsize.py tests/test_gpu_fullsize.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt
    tail -3 $O/pytest.log
    timeout 900 bash tools/gap_report.sh r06g > $O/gap_report.log 2>&1; echo "gap report rc=$?" | tee -a $O/status.txt
    head -22 gpurun_out/gap_r06g/report.txt | cut -c1-200; grep -A12 "largest single" gpurun_out/gap_r06g/report.txt | cut -c1-200
    timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 --no-configs --no-shards --no-extras > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc=$?" | tee -a $O/status.txt
    python -c "
import json; j=json.loads(open('$O/bench_short.json').read().strip().splitlines()[-1]); print('bench', j['value'], j['ms_per_step'], j['roofline']['frac'])"
}

# round 6, GPU call 8: same-box A/B of the host-side change (device-built row tables, vectorised packing): 3 timed steps each, alternating, against the
# tree of the commit before it (tools/ab/base_tree: same library, old s2mel.py / codec.py)
call8() {
    O=$PWD/gpurun_out/r06h
    mkdir -p $O
    for rep in 1 2; do
        (cd tools/ab/base_tree && timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 --no-configs --no-shards --no-extras) > $O/base_$rep.json 2> $O/base_$rep.err
        timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 --no-configs --no-shards --no-extras > $O/new_$rep.json 2> $O/new_$rep.err
    done
    python - <<PY
import json
for n in ("base_1", "new_1", "base_2", "new_2"):
    try:
        j = json.loads(open("$O/" + n + ".json").read().strip().splitlines()[-1]); print(n, round(j["value"], 2), round(j["ms_per_step"], 1), round(j["roofline"]["frac"], 4))
    except Exception as e:
        print(n, "failed", e)
PY
}

# round 6, GPU call 9: PMC refresh at the round's code (separate --pmc passes, --kernel-trace only): matrix-pipe busy / clock of the x3 kernels, HBM
# traffic of the x3 GEMM at the bench shape, the GPT kernels at 64 rows, the vocoder conv traffic
call9() {
    O=$PWD/gpurun_out/r06i
    mkdir -p $O
    timeout 900 bash tools/pmc_x3.sh 8 > $O/pmc_x3.log 2>&1; echo "pmc_x3 rc=$?" | tee $O/status.txt; cp gpurun_out/pmc_x3/x3_pmc.json $O/ 2>/dev/null; tail -12 $O/pmc_x3.log | cut -c1-300
    timeout 900 bash tools/pmc_s2mel_traffic.sh 64 fp32x3 > $O/pmc_s2mel.log 2>&1; echo "pmc_s2mel rc=$?" | tee -a $O/status.txt; tail -2 $O/pmc_s2mel.log | cut -c1-600
    cp gpurun_out/pmc_s2mel/s2mel_gemm_traffic.json $O/ 2>/dev/null
    timeout 900 bash tools/pmc_gpt.sh 64 24 > $O/pmc_gpt.log 2>&1; echo "pmc_gpt b64 rc=$?" | tee -a $O/status.txt; tail -3 $O/pmc_gpt.log | cut -c1-800
    cp gpurun_out/pmc_gpt/gpt_pmc_b64.json $O/ 2>/dev/null
}

# round 6, GPU call 10: x3 GEMM model with EIGHT waves per 128 x 128 block (32 x 64 wave tiles, 123 registers: four waves per SIMD at two blocks per
# CU) against the product structure, with its own ablations (tools/microbench/x3_gemm_lab4.hip)
call10() {
    O=$PWD/gpurun_out/r06j
    mkdir -p $O
    timeout 600 tools/microbench/bin/x3_gemm_lab4 > $O/x3_gemm_lab4.log 2>&1; echo "x3_gemm_lab4 rc=$?" | tee $O/status.txt
    cat $O/x3_gemm_lab4.log | cut -c1-170
}

# round 6, GPU call 11: the 8-wave x3 GEMM in the product (option x3_waves = 8): bitwise tests (GEMM shapes, estimator, run-to-run), GEMM shapes and
# the 64-utterance solve against the 4-wave kernel, same box, alternating
call11() {
    O=$PWD/gpurun_out/r06k
    mkdir -p $O
    timeout 1500 python -m pytest tests/test_gpu_gemm_x3.py tests/test_gpu_s2mel.py -x -q > $O/pytest_x3.log 2>&1; echo "pytest x3 / s2mel rc=$?" | tee $O/status.txt
    tail -4 $O/pytest_x3.log
    for wv in 4 8 4 8; do
        timeout 300 python tools/gemm_x3_bench.py 312704 5 x3_waves=$wv > $O/gemm_w$wv.log 2>&1; echo "x3_waves=$wv: $(grep 'f32x3' $O/gemm_w$wv.log | sed 's/.*f32x3: //' | awk '{printf "%s ", $3}')"
    done
    timeout 900 python tools/s2mel_bench.py 64 517 1926 1 fp32x3 fp32x3:x3_waves=8 fp32x3:x3_waves=4 fp32x3:x3_waves=8 fp32x3:x3_waves=4 fp32x3:x3_waves=8 > $O/solve_waves.log 2>&1; echo "solve rc=$?" | tee -a $O/status.txt
    grep "^B=" $O/solve_waves.log | cut -c1-260
}

# round 6, GPU call 12: 8 waves per block as the x3 GEMM's default: s2mel / x3 / full-size tests, PMC of the new kernel (matrix-pipe busy, HBM traffic)
call12() {
    O=$PWD/gpurun_out/r06l
    mkdir -p $O
    timeout 1800 python -m pytest tests/test_gpu_gemm_x3.py tests/test_gpu_s2mel.py tests/test_gpu_attn_x3.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline_fullsize.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt
    tail -3 $O/pytest.log
    timeout 900 bash tools/pmc_x3.sh 8 > $O/pmc_x3.log 2>&1; echo "pmc_x3 rc=$?" | tee -a $O/status.txt; cp gpurun_out/pmc_x3/x3_pmc.json $O/ 2>/dev/null
    python -c "
import json; j=json.load(open('$O/x3_pmc.json'))
for k,v in j.items():
    if isinstance(v,dict): print(k, round(v['mfma_busy_of_own_cycles'],3), round(v['effective_clock_GHz'],3), v['dispatches'])"
    timeout 900 bash tools/pmc_s2mel_traffic.sh 64 fp32x3 > $O/pmc_s2mel.log 2>&1; echo "pmc_s2mel rc=$?" | tee -a $O/status.txt; tail -1 $O/pmc_s2mel.log | cut -c1-400
    cp gpurun_out/pmc_s2mel/s2mel_gemm_traffic.json $O/ 2>/dev/null
    timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --alt-steps 0 --no-configs --no-shards --no-extras > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc=$?" | tee -a $O/status.txt
    python -c "
import json; j=json.loads(open('$O/bench_short.json').read().strip().splitlines()[-1]); print('bench', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['achieved'])"
}

# round 6, GPU call 13: how fast does one block per CU fill its 160 KiB slab by LDS-DMA with 4 / 8 / 16 issuing waves (tools/microbench/slab_fill.hip)
call13() {
    O=$PWD/gpurun_out/r06o
    mkdir -p $O
    timeout 300 tools/microbench/bin/slab_fill > $O/slab_fill.log 2>&1; echo "slab_fill rc=$?" | tee $O/status.txt
    cat $O/slab_fill.log
}

# round 6, GPU call 14: is the x3 vocoder conv at the power limit (clock ~2.0 GHz) or stalled below it?  matrix-pipe busy + clock per vocoder kernel
call14() {
    O=$PWD/gpurun_out/r06p
    mkdir -p $O
    timeout 900 bash tools/pmc_voc_x3.sh 16 > $O/pmc_voc_x3.log 2>&1; echo "pmc_voc_x3 rc=$?" | tee $O/status.txt
    cp gpurun_out/pmc_voc_x3/voc_x3_pmc.json $O/ 2>/dev/null; cat $O/pmc_voc_x3.log | tail -45
}

# round 6, GPU call 15: the x3 vocoder conv with sixteen waves per block (option voc_x3_waves = 16): bitwise tests, forward + per-stage conv rates at
# 16 x 1926 frames against the 8-wave kernel, alternating
call15() {
    O=$PWD/gpurun_out/r06r
    mkdir -p $O
    timeout 900 python -m pytest tests/test_gpu_bigvgan_x3.py -x -q > $O/pytest_voc_x3.log 2>&1; echo "pytest voc x3 rc=$?" | tee $O/status.txt; tail -3 $O/pytest_voc_x3.log
    timeout 900 python tools/voc_h3_bench.py 16 bf16x3:96,bf16x3:96:voc_x3_waves=16,bf16x3:96,bf16x3:96:voc_x3_waves=16 > $O/voc_bench.log 2>&1; echo "voc bench rc=$?" | tee -a $O/status.txt
    grep -v amdgpu.ids $O/voc_bench.log | cut -c1-150
}

# round 6, GPU call 15 (ran at commit "x3 vocoder conv with sixteen waves per block"; the variant was removed after it): pytest tests/test_gpu_bigvgan_x3.py
# with the 16-vs-8-wave bitwise test, then  tools/voc_h3_bench.py 16 bf16x3:96,bf16x3:96:voc_x3_waves=16,bf16x3:96,bf16x3:96:voc_x3_waves=16  -> profiles/r06r

"call$1"
