#!/bin/bash
# GPU box: three tile GEMM kernels A/B (bitwise + per-kernel time), bf16 shadow outputs instead of cast passes.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_gpt.py -m gpu -q -s -k "tile or s2mel" > $O/pytest_tile.log 2>&1; echo "pytest_tile rc=$?" > $O/status.txt
for v in 0 1 2; do
  ITTS_TILE256=$v timeout 200 python tools/prefill_bench.py 2>&1 | grep TFLOP | sed "s/^/TILE256=$v /" >> $O/prefill_bench.log
  ITTS_TILE256=$v timeout 300 python tools/s2mel_bench.py 8 800 1926 25 bf16 2>&1 | grep "ms total" | sed "s/^/TILE256=$v /" >> $O/s2mel_bench.log
  ITTS_TILE256=$v ITTS_FA_QS=2 timeout 300 python tools/s2mel_bench.py 32 517 1926 25 bf16 2>&1 | grep "ms total" | sed "s/^/TILE256=$v QS=2 /" >> $O/s2mel_bench.log
done
cd /tmp && export TMPDIR=/tmp
ITTS_TILE256=2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s2mel_2 -o s -- python $GRAFT_REPO_ROOT/tools/s2mel_bench.py 32 517 1926 3 bf16 > $GRAFT_REPO_ROOT/$O/s2mel_prof_2.log 2>&1
f=$(find /tmp/prof_s2mel_2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/s2mel_kernel_stats_b32_tile256_2.csv
cd $GRAFT_REPO_ROOT
cat $O/status.txt; tail -3 $O/pytest_tile.log; cat $O/prefill_bench.log; cat $O/s2mel_bench.log
