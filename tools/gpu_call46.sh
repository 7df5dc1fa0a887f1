#!/bin/bash
# round 3, call 12: dead-row elimination (bit-identity test) and the f32 flash kernel with two query sub-tiles per wave: tests, then A/B timings.
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03m
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_gemm_x3.py -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/status.txt
timeout 200 python tools/s2mel_bench.py 64 517 1926 3 fp32 2>&1 | tail -1 | sed 's/^/QS=2, prune: /' >> $O/s2mel.log
ITTS_FA32_QS=1 timeout 200 python tools/s2mel_bench.py 64 517 1926 3 fp32 2>&1 | tail -1 | sed 's/^/QS=1, prune: /' >> $O/s2mel.log
ITTS_S2MEL_PRUNE=0 timeout 200 python tools/s2mel_bench.py 64 517 1926 3 fp32 2>&1 | tail -1 | sed 's/^/QS=2, no prune: /' >> $O/s2mel.log
timeout 200 python tools/s2mel_bench.py 64 517 1926 5 bf16 2>&1 | tail -1 | sed 's/^/bf16 prune: /' >> $O/s2mel.log
ITTS_S2MEL_PRUNE=0 timeout 200 python tools/s2mel_bench.py 64 517 1926 5 bf16 2>&1 | tail -1 | sed 's/^/bf16 no prune: /' >> $O/s2mel.log
cat $O/status.txt; grep -E "passed|failed|rror|FLOPs|attention unit|max\|d\|" $O/pytest.log | tail -14 | cut -c1-260; cat $O/s2mel.log
