#!/bin/bash
# round 3, call 22: LayerNorm-fused decode GEMMs after pinning the contraction of ln_row: the GPT / compaction / pipeline GPU tests
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03v
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_compaction.py tests/test_gpu_edges.py tests/test_gpu_pipeline.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/status.txt
cat $O/status.txt; tail -6 $O/pytest.log | cut -c1-300
