#!/bin/bash
# round 3, call 5: f32x3 GEMM after the LDS fix (two blocks per CU), row compaction + per-row caps, then the timings.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03e
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm_x3.py tests/test_gpu_compaction.py -x -q -s > $O/pytest_a.log 2>&1; echo "pytest x3+compaction rc=$?" > $O/status.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s > $O/pytest_b.log 2>&1; echo "pytest fullsize rc=$?" >> $O/status.txt
for prec in fp32x3; do timeout 200 python tools/s2mel_bench.py 8 517 1926 3 $prec 2>&1 | tail -1 >> $O/s2mel_steps.log; done
ITTS_X3_PRODUCTS=6 timeout 200 python tools/s2mel_bench.py 8 517 1926 3 fp32x3 2>&1 | tail -1 | sed 's/^/6-product: /' >> $O/s2mel_steps.log
timeout 900 python bench.py --steps 1 --warmup 1 --s2mel-precision fp32x3 --alt-steps 0 --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3.err; echo "bench rc=$?" >> $O/status.txt
cat $O/status.txt; grep -E "rms|max\|d\||max-rel|passed|failed|rror|row-steps|margin" $O/pytest_a.log $O/pytest_b.log | tail -40 | cut -c1-400; cat $O/s2mel_steps.log; tail -4 $O/bench_x3.err | cut -c1-600
