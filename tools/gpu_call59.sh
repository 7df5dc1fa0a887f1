#!/bin/bash
# round 3, call 25: sampler / beam kernel prologue with every row load issued up front: decode ms/token, then the GPT-side GPU tests and the
# suites that run on itts_layernorm_forward (ln_row's contraction was pinned)
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03y
mkdir -p $O
for B in 1 64; do
  ITTS_BEAM_BENCH_MODES=2 timeout 200 python tools/beam_bench.py $B 200 2>&1 | grep "^B=" | cut -c1-120 >> $O/sample_prefetch.log
done
timeout 300 python tools/config0_check.py 2>&1 | grep "^bf16" | cut -c1-330 >> $O/sample_prefetch.log
timeout 1200 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_compaction.py tests/test_gpu_pipeline.py tests/test_gpu_cond.py tests/test_gpu_codec.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/status.txt
cat $O/status.txt; tail -3 $O/pytest.log | cut -c1-300; cat $O/sample_prefetch.log
