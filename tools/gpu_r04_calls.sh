#!/bin/bash
# The exact command lists of round 4's GPU sessions (each ran on an MI355X box through gpurun; outputs under gpurun_out/, the summaries that are
# evidence were copied to profiles/r04*).  usage: tools/gpu_r04_calls.sh <n>      e.g.  gpurun --timeout 2400 -- 'bash tools/gpu_r04_calls.sh 13'
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD

# round 4, GPU call 1: GPT tests on the new attention / fused-LN kernels, decode ms/token by batch and option, one decode-step timeline at 8
# rows, the repaired GPT PMC tool at 8 rows.
call1() {
    O=$PWD/gpurun_out/r04a
    mkdir -p $O
    timeout 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_compaction.py tests/test_gpu_edges.py -x -q > $O/pytest_gpt.log 2>&1; echo "pytest gpt rc=$?" | tee $O/status.txt; tail -5 $O/pytest_gpt.log
    timeout 600 python tools/decode_bench.py 560 1,4,8,16,32,64 attn_waves=4 attn_waves=8 attn_waves=16 decode_fuse_ln=0 > $O/decode_bench.log 2>&1; echo "decode_bench rc=$?" | tee -a $O/status.txt
    cat $O/decode_bench.log | grep "^B="
    timeout 300 bash tools/trace_decode.sh 8 400 > $O/trace8.log 2>&1; cp gpurun_out/trace_decode/step_timeline.txt $O/decode_step_timeline_b8.txt 2>/dev/null; head -14 $O/decode_step_timeline_b8.txt; tail -2 $O/decode_step_timeline_b8.txt
    timeout 600 bash tools/pmc_gpt.sh 8 24 > $O/pmc_gpt_b8.log 2>&1; echo "pmc_gpt b8 rc=$?" | tee -a $O/status.txt; cp gpurun_out/pmc_gpt/gpt_pmc_b8.json $O/ 2>/dev/null; tail -30 $O/pmc_gpt_b8.log | head -60
}

# round 4, GPU call 2: weight-prefetch microbenchmark (does touching the next kernel's weights shorten the LN -> GEMM chain?), the 6-product
# x3 GEMM against an f64 product on every s2mel shape, the three s2mel modes against the reference classes' production-width fixture.
call2() {
    O=$PWD/gpurun_out/r04b
    mkdir -p $O
    timeout 150 tools/microbench/bin/weight_prefetch > $O/weight_prefetch.log 2>&1; echo "weight_prefetch rc=$?" | tee $O/status.txt
    cat $O/weight_prefetch.log
    timeout 600 python -m pytest tests/test_gpu_gemm_x3.py -x -q -s > $O/pytest_x3.log 2>&1; echo "pytest x3 rc=$?" | tee -a $O/status.txt
    grep -E "GEMM|variant|passed|failed|Error" $O/pytest_x3.log | tail -30
    timeout 600 python -m pytest tests/test_gpu_s2mel.py -x -q -s -k "production" > $O/pytest_s2mel_prod.log 2>&1; echo "pytest s2mel prod rc=$?" | tee -a $O/status.txt
    grep -E "production|passed|failed|Error" $O/pytest_s2mel_prod.log | tail -12
}

# round 4, GPU call 3: the fp32x3 mode's attention on bf16 planes (flash_attn_x3_kernel): unit tests against an f64 attention and the native f32
# flash kernel, the s2mel suite (every mode against the reference classes' fixtures), one solve at the bench's per-utterance shape per mode.
call3() {
    O=$PWD/gpurun_out/r04c
    mkdir -p $O
    timeout 300 python -m pytest tests/test_gpu_attn_x3.py -x -q -s > $O/pytest_attn_x3.log 2>&1; echo "pytest attn_x3 rc=$?" | tee $O/status.txt
    grep -E "attention|passed|failed|Error|error" $O/pytest_attn_x3.log | tail -12
    timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_gemm_x3.py -x -q -s > $O/pytest_s2mel.log 2>&1; echo "pytest s2mel rc=$?" | tee -a $O/status.txt
    grep -E "production|passed|failed|Error" $O/pytest_s2mel.log | tail -12
    timeout 600 python tools/s2mel_bench.py 8 517 1926 5 fp32 fp32x3:x3_attn=0,x3_products=8 fp32x3:x3_attn=0,x3_products=6 fp32x3:x3_products=8 fp32x3:x3_products=6 > $O/s2mel_bench.log 2>&1; echo "s2mel_bench rc=$?" | tee -a $O/status.txt
    grep "^B=" $O/s2mel_bench.log
}

# round 4, GPU call 4: why does the 8-product x3 solve sit 2-4e-3 from the f32 solve at 8 x 2443 frames when the 6-product one sits at 8e-6?
call4() {
    O=$PWD/gpurun_out/r04d
    mkdir -p $O
    timeout 300 python tools/s2mel_bench.py 8 517 1926 1 fp32 fp32x3:x3_attn=0,x3_products=8 fp32x3:x3_attn=0,x3_products=8,x3_sched=0 fp32x3:x3_attn=0,x3_products=8 fp32x3:x3_attn=0,x3_products=6,x3_sched=0 > $O/dbg_b8.log 2>&1; echo "b8 rc=$?"
    grep "^B=" $O/dbg_b8.log
    timeout 300 python tools/s2mel_bench.py 2 517 1926 1 fp32 fp32x3:x3_attn=0,x3_products=8 fp32x3:x3_attn=0,x3_products=8,x3_sched=0 fp32x3:x3_attn=0,x3_products=6 > $O/dbg_b2.log 2>&1; echo "b2 rc=$?"
    grep "^B=" $O/dbg_b2.log
    timeout 300 python - > $O/dbg_gemm.log 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, ".")
from indextts_amd import _lib, gpt
g = torch.Generator().manual_seed(5)
for (M, N, K) in [(39088, 1536, 512), (39088, 512, 1536), (39088, 512, 512), (9720, 1536, 512)]:
    a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    ref = (a.double() @ w.double().t() + b.double())
    for prec, prods in (("fp32", None), ("fp32x3", 8), ("fp32x3", 6)):
        with _lib.option_scope(**({"x3_products": prods} if prods else {})):
            wp = gpt.pack_gemm_weight(w, prec).to("cuda:0")
            y = gpt.gemm(a.to("cuda:0"), wp, b.to("cuda:0"), N, prec, prefill_tiles=True).cpu().double()
        d = y - ref
        bad = (d.abs() > 1e-4).nonzero()
        print(f"GEMM {M}x{N}x{K} {prec} {prods}: max|d| {float(d.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e} bad {len(bad)} first {bad[:3].tolist()}", flush=True)
PY
    echo "gemm rc=$?"; cat $O/dbg_gemm.log | tail -14
}

# round 4, GPU call 5: the new full-size parity tests (3-beam beam-sample ids vs the reference's, pipeline level at configs[1] / configs[4] sizes vs
# the oracle chain), the s2mel + x3 suites with the determinism test, a short bench line in the new default mode.
call5() {
    O=$PWD/gpurun_out/r04e
    mkdir -p $O
    timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "beam_sample" > $O/pytest_fullsize_beam.log 2>&1; echo "pytest fullsize beam rc=$?" | tee $O/status.txt
    grep -E "GPT 24|passed|failed|Error|row " $O/pytest_fullsize_beam.log | tail -5
    timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_gemm_x3.py tests/test_gpu_attn_x3.py -x -q -s > $O/pytest_s2mel.log 2>&1; echo "pytest s2mel rc=$?" | tee -a $O/status.txt
    grep -E "bits|passed|failed|Error" $O/pytest_s2mel.log | tail -8
    timeout 1500 python -m pytest tests/test_gpu_pipeline_fullsize.py -x -q -s > $O/pytest_pipeline_fullsize.log 2>&1; echo "pytest pipeline fullsize rc=$?" | tee -a $O/status.txt
    grep -E "configs\[|passed|failed|Error" $O/pytest_pipeline_fullsize.log | tail -12
    timeout 900 python bench.py --steps 2 --warmup 1 --no-configs --no-shards --no-cpu-baseline --alt-steps 1 > $O/bench_short.json 2> $O/bench_short.log; echo "bench rc=$?" | tee -a $O/status.txt
    python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r04e/bench_short.json"))
    print("value", j["value"], "ms_per_step", j["ms_per_step"], "roofline", {k: j["roofline"][k] for k in ("achieved", "peak", "frac", "ms_per_step")})
    print("s2mel", {k: v for k, v in j["stages"]["s2mel"].items() if "ms_per_step" in k or "tflops" in k})
    print("by precision", j.get("value_by_s2mel_precision"))
    print("gpt", j["stages"]["gpt_decode_ms_per_token"], "bigvgan", j["stages"]["bigvgan_ms_per_step"])
except Exception as e:
    print("bench json:", repr(e))
PY
    tail -5 $O/bench_short.log
}

# round 4, GPU call 6: the full-size parity tests added last (3-beam beam-sample ids vs the reference's, pipeline level at configs[1] /
# configs[4] sizes vs the oracle chain), the bench line in the new default mode (fp32x3, 6 plane products, x3 attention) with the per-rank
# shards, and the rocprofv3 kernel stats of the same command.
call6() {
    O=$ROOT/gpurun_out/r04e
    mkdir -p $O
    timeout 400 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "beam_sample" > $O/pytest_fullsize_beam.log 2>&1; echo "pytest fullsize beam rc=$?" | tee $O/status.txt
    grep -E "GPT 24|passed|failed|Error|row " $O/pytest_fullsize_beam.log | tail -5
    timeout 900 python -m pytest tests/test_gpu_pipeline_fullsize.py -x -q -s > $O/pytest_pipeline_fullsize.log 2>&1; echo "pytest pipeline fullsize rc=$?" | tee -a $O/status.txt
    grep -E "configs\[|passed|failed|Error" $O/pytest_pipeline_fullsize.log | tail -12
    timeout 900 python bench.py --steps 3 --warmup 1 --no-configs > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/status.txt
    python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r04e/bench.json"))
    print("value", j["value"], "ms_per_step", j["ms_per_step"], "dtype", j["dtype"], "roofline", j["roofline"])
    print("cpu_baseline", j["cpu_baseline"])
    st = j["stages"]
    print({k: v for k, v in st.items() if not isinstance(v, (dict, list))})
    print("by precision", j.get("value_by_s2mel_precision"))
    print({k: v for k, v in st.get("configs", {}).items()})
except Exception as e:
    print("bench json:", repr(e))
PY
    tail -3 $O/bench.log
    cd /tmp && export TMPDIR=/tmp
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --no-configs --no-shards --no-cpu-baseline --alt-steps 0 --no-extras > $O/bench_prof.json 2> $O/bench_prof.err; echo "rocprof rc=$?" | tee -a $O/status.txt
    find $O/raw -name "*kernel_stats*.csv" -exec cp {} $O/bench_kernel_stats.csv \; 2>/dev/null
    rm -rf $O/raw
    head -12 $O/bench_kernel_stats.csv | cut -c1-200
}

# round 4, GPU call 7: the three-stage A ring of the x3 GEMM (option x3_stages = 3) -- bitwise test against the two-stage kernel, then the
# 8 x 2443-frame solve in alternating pairs, under rocprofv3 so that the per-kernel averages of both variants come from one process.
call7() {
    O=$ROOT/gpurun_out/r04f
    mkdir -p $O
    timeout 300 python -m pytest tests/test_gpu_gemm_x3.py -x -q -s -k "three_stage" > $O/pytest_ring3.log 2>&1; echo "pytest ring3 rc=$?" | tee $O/status.txt
    tail -3 $O/pytest_ring3.log
    timeout 300 python tools/s2mel_bench.py 8 517 1926 3 fp32x3 fp32x3:x3_stages=3 fp32x3 fp32x3:x3_stages=3 > $O/s2mel_bench.log 2>&1; echo "s2mel_bench rc=$?" | tee -a $O/status.txt
    grep "^B=" $O/s2mel_bench.log
    cd /tmp && export TMPDIR=/tmp
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o sb -- python $ROOT/tools/s2mel_bench.py 8 517 1926 2 fp32x3 fp32x3:x3_stages=3 > $O/s2mel_bench_prof.log 2>&1; echo "rocprof rc=$?" | tee -a $O/status.txt
    find $O/raw -name "*kernel_stats*.csv" -exec cp {} $O/s2mel_kernel_stats.csv \; 2>/dev/null
    rm -rf $O/raw
    grep "gemm_x3" $O/s2mel_kernel_stats.csv | cut -c1-160
}

# round 4, GPU call 8: vmcnt retirement order across VGPR loads / LDS-DMA (microbenchmark), the configs[3]-size pipeline parity test, and the
# PMC passes behind the bench line's traffic figures (x3 GEMM, vocoder conv) and the GPT decode / prefill counters at 64 rows.
call8() {
    O=$ROOT/gpurun_out/r04g
    mkdir -p $O
    timeout 120 tools/microbench/bin/vmcnt_order > $O/vmcnt_order.log 2>&1; echo "vmcnt_order rc=$?" | tee $O/status.txt
    cat $O/vmcnt_order.log
    timeout 900 python -m pytest tests/test_gpu_pipeline_fullsize.py -x -q -s -k config3 > $O/pytest_config3.log 2>&1; echo "pytest config3 rc=$?" | tee -a $O/status.txt
    grep -E "configs\[|passed|failed|Error|error" $O/pytest_config3.log | tail -8
    timeout 600 bash tools/pmc_s2mel_traffic.sh 64 fp32x3 > $O/pmc_s2mel.log 2>&1; echo "pmc_s2mel rc=$?" | tee -a $O/status.txt
    tail -2 $O/pmc_s2mel.log | cut -c1-600
    cp gpurun_out/pmc_s2mel/s2mel_gemm_traffic.json $O/ 2>/dev/null
    timeout 900 bash tools/pmc_gpt.sh 64 24 > $O/pmc_gpt.log 2>&1; echo "pmc_gpt b64 rc=$?" | tee -a $O/status.txt
    tail -3 $O/pmc_gpt.log | cut -c1-800
    cp gpurun_out/pmc_gpt/gpt_pmc_b64.json $O/ 2>/dev/null
    timeout 600 bash tools/pmc_bench_traffic.sh 64 > $O/pmc_conv.log 2>&1; echo "pmc_conv rc=$?" | tee -a $O/status.txt
    tail -2 $O/pmc_conv.log | cut -c1-600
    ls gpurun_out/pmc_bench/ | head; cp gpurun_out/pmc_bench/*.json $O/ 2>/dev/null
}

# round 4, GPU call 9: the wide LayerNorm-fused decode GEMM (5-16 rows: weights on waves 0-3, LayerNorm on waves 4-7, 2 / 4 n-tiles per block) --
# bitwise tests against the two launches it replaces, whole decode loops, then ms / token at 5-16 rows against the separate-LayerNorm path.
call9() {
    O=$PWD/gpurun_out/r04h
    mkdir -p $O
    timeout 600 python -m pytest tests/test_gpu_gpt.py -x -q -k "layernorm_fused or fused_layernorm" > $O/pytest_lnw.log 2>&1; echo "pytest lnw rc=$?" | tee $O/status.txt
    tail -4 $O/pytest_lnw.log
    timeout 600 python tools/decode_bench.py 560 5,8,12,16 decode_fuse_ln=2 decode_fuse_ln=2,decode_ln_nt=2 decode_fuse_ln=2,decode_ln_nt=0 > $O/decode_bench.log 2>&1; echo "decode_bench rc=$?" | tee -a $O/status.txt
    grep "^B=" $O/decode_bench.log
}

# round 4, GPU call 10: the MFMA -> inline-asm hazard microbenchmark, then -- with fa_max3 as compiler-generated code -- the attention / s2mel suites,
# the fused-LayerNorm decode tests at the new default (1-8 rows), and run-to-run determinism of one estimator call per mode (16 repetitions).
call10() {
    O=$PWD/gpurun_out/r04i
    mkdir -p $O
    timeout 120 tools/microbench/bin/mfma_asm_hazard > $O/mfma_asm_hazard.log 2>&1; echo "mfma_asm_hazard rc=$?" | tee $O/status.txt
    cat $O/mfma_asm_hazard.log
    SOLVE=0 timeout 600 python tools/s2mel_determinism.py 2 517 1926 1 16 bf16 fp32x3 fp32 bf16:tile256=0 > $O/determinism_b2.log 2>&1; echo "determinism rc=$?" | tee -a $O/status.txt
    grep -E "^poison|first bad" $O/determinism_b2.log | cut -c1-900
    timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_attn_x3.py tests/test_gpu_gemm_x3.py -x -q > $O/pytest_s2mel.log 2>&1; echo "pytest s2mel rc=$?" | tee -a $O/status.txt
    tail -3 $O/pytest_s2mel.log
    timeout 900 python -m pytest tests/test_gpu_gpt.py -x -q -k "layernorm_fused or fused_layernorm or invarian or compaction" > $O/pytest_gpt_ln.log 2>&1; echo "pytest gpt ln rc=$?" | tee -a $O/status.txt
    tail -3 $O/pytest_gpt_ln.log
}

# round 4, GPU call 11: which stage of the bf16 s2mel estimator is not bit-stable?  (engine trace checksums, 24 repetitions per setting)
call11() {
    O=$PWD/gpurun_out/r04j
    mkdir -p $O
    timeout 600 python tools/s2mel_trace.py 2 517 1926 24 bf16 bf16:s2mel_fused=0 bf16:tile256=0 fp32x3 > $O/trace_b2.log 2>&1; echo "trace rc=$?" | tee $O/status.txt
    grep -v "amdgpu.ids" $O/trace_b2.log | cut -c1-400
}

call12() {
    O=$PWD/gpurun_out/r04k
    mkdir -p $O
    timeout 600 python tools/s2mel_trace.py 2 517 1926 24 bf16 bf16:dbg=1 bf16:dbg=2 bf16:dbg=4 bf16:dbg=8 > $O/trace_dbg.log 2>&1; echo "trace rc=$?" | tee $O/status.txt
    grep -v "amdgpu.ids" $O/trace_dbg.log | grep -v "repetition" | cut -c1-600
    DEPTH=1 WN_LAYERS=1 timeout 300 python tools/s2mel_trace.py 2 517 1926 200 bf16 > $O/trace_depth1.log 2>&1; echo "trace depth1 rc=$?" | tee -a $O/status.txt
    grep -v "amdgpu.ids" $O/trace_depth1.log | grep -v "repetition" | cut -c1-600
}

# round 4, GPU call 13: the round's full validation -- every GPU test, smoke, the bench line as the driver runs it, the rocprofv3 kernel stats of
# the same command, and the bf16 stage trace with the wqkv fusion off (default) / on.
call13() {
    O=$ROOT/gpurun_out/r04m
    mkdir -p $O
    timeout 300 python tools/s2mel_trace.py 2 517 1926 24 bf16 bf16:s2mel_fused=2 > $O/trace_bf16.log 2>&1; echo "trace rc=$?" | tee $O/status.txt
    grep -v "amdgpu.ids" $O/trace_bf16.log | grep -v "   repetition" | cut -c1-400
    timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/status.txt
    tail -5 $O/pytest_gpu.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.txt
    tail -2 $O/smoke.log
    timeout 1200 python bench.py > $O/bench.json 2> $O/bench.log; echo "bench rc=$?" | tee -a $O/status.txt
    python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r04m/bench.json"))
    print("value", j["value"], "ms_per_step", j["ms_per_step"], "roofline", {k: j["roofline"][k] for k in ("achieved", "peak", "frac", "traffic", "ms_per_step")})
    print("by precision", j.get("value_by_s2mel_precision"))
    st = j["stages"]
    print({k: v for k, v in st.items() if not isinstance(v, (dict, list))})
    print({k: (v.get("audio_seconds_per_sec"), v.get("ms_per_step")) for k, v in st.get("configs", {}).items() if isinstance(v, dict)})
    print("cpu", j["cpu_baseline"]["value"])
except Exception as e:
    print("bench json:", repr(e))
PY
    tail -3 $O/bench.log
    cd /tmp && export TMPDIR=/tmp
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o bench -- python $ROOT/bench.py --steps 1 --warmup 1 --no-configs --no-shards --no-cpu-baseline --alt-steps 0 --no-extras > $O/bench_prof.json 2> $O/bench_prof.err; echo "rocprof rc=$?" | tee -a $O/status.txt
    find $O/raw -name "*kernel_stats*.csv" -exec cp {} $O/bench_kernel_stats.csv \; 2>/dev/null
    rm -rf $O/raw
    head -8 $O/bench_kernel_stats.csv | cut -c1-160
}

# round 4, GPU call 14: 32-row decode GEMM blocks (80 KiB slab, two blocks per CU) against the 64-row form above 32 rows
call14() {
    O=$PWD/gpurun_out/r04n
    mkdir -p $O
    timeout 600 python tools/decode_bench.py 560 40,48,64 decode_mt=2 decode_mt=2,decode_nt=0 > $O/decode_bench.log 2>&1; echo "decode_bench rc=$?" | tee $O/status.txt
    grep "^B=" $O/decode_bench.log
}

# round 4, GPU call 15: n-tiles per block of the 64-row decode GEMM (slab re-reads from L2 vs blocks streaming weights)
call15() {
    O=$PWD/gpurun_out/r04o
    mkdir -p $O
    timeout 600 python tools/decode_bench.py 560 32,64 decode_nt=1 decode_nt=2 decode_nt=4 > $O/decode_bench.log 2>&1; echo "decode_bench rc=$?" | tee $O/status.txt
    grep "^B=" $O/decode_bench.log
}

case "${1:-}" in
    1) call1 ;;
    2) call2 ;;
    3) call3 ;;
    4) call4 ;;
    5) call5 ;;
    6) call6 ;;
    7) call7 ;;
    8) call8 ;;
    9) call9 ;;
    10) call10 ;;
    11) call11 ;;
    12) call12 ;;
    13) call13 ;;
    14) call14 ;;
    15) call15 ;;
    *) echo "usage: $0 <1|2|3|4|5|6|7|8|9|10|11|12|13|14|15>"; exit 2 ;;
esac
