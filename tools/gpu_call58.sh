#!/bin/bash
# round 3, call 24: ballot-bisection top-k in the sampling / beam kernels: GPT + compaction + pipeline tests, then decode ms/token A/B
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03x
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_compaction.py tests/test_gpu_pipeline.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/status.txt
for B in 1 64; do for f in 1 0 1 0; do
  ITTS_SAMPLE_RADIX=$f ITTS_BEAM_BENCH_MODES=2 timeout 200 python tools/beam_bench.py $B 200 2>&1 | grep "^B=" | cut -c1-120 | sed "s/^/radix=$f /" >> $O/sample_topk.log
done; done
cat $O/status.txt; tail -4 $O/pytest.log | cut -c1-300; cat $O/sample_topk.log
