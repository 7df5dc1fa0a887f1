#!/bin/bash
# round 3, call 4: the f32x3 GEMM (f32 operands as three bf16 planes on the bf16 MFMA) -- accuracy vs f64 beside the native f32 kernel,
# the s2mel solve in that mode vs the reference goldens and at the bench's size (mel / waveform error after 25 steps, with bf16 beside
# it), then what it does to the step: solve timing at B = 8 and a 1-step bench in that mode.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm_x3.py tests/test_gpu_fullsize.py -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/status.txt
for prec in fp32 fp32x3; do timeout 200 python tools/s2mel_bench.py 8 517 1926 3 $prec 2>&1 | tail -1 >> $O/s2mel_steps.log; done
ITTS_X3_PRODUCTS=6 timeout 200 python tools/s2mel_bench.py 8 517 1926 3 fp32x3 2>&1 | tail -1 | sed 's/^/6-product: /' >> $O/s2mel_steps.log
timeout 900 python bench.py --steps 1 --warmup 1 --s2mel-precision fp32x3 --alt-steps 0 --no-configs --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err; echo "bench rc=$?" >> $O/status.txt
cat $O/status.txt; grep -E "rms|max\|d\||max-rel|passed|failed|rror" $O/pytest.log | tail -40; cat $O/s2mel_steps.log; tail -3 $O/bench_x3.err; cut -c1-400 $O/bench_x3.json
