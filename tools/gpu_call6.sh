#!/bin/bash
# GPU box: 256x256 eight-wave tile GEMM vs the 128x128 kernel (bitwise A/B + speed), transposed V^T store.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_gpt.py -m gpu -q -s -k "tile or s2mel or prefill" > $O/pytest_tile.log 2>&1; echo "pytest_tile rc=$?" > $O/status.txt
for v in 0 1; do
  ITTS_TILE256=$v timeout 200 python tools/prefill_bench.py 2>&1 | grep TFLOP | sed "s/^/TILE256=$v /" >> $O/prefill_bench.log
  ITTS_TILE256=$v timeout 300 python tools/s2mel_bench.py 8 800 1926 25 bf16 2>&1 | grep "ms total" | sed "s/^/TILE256=$v /" >> $O/s2mel_bench.log
  ITTS_TILE256=$v timeout 300 python tools/s2mel_bench.py 32 517 1926 25 bf16 2>&1 | grep "ms total" | sed "s/^/TILE256=$v /" >> $O/s2mel_bench.log
done
cd /tmp && export TMPDIR=/tmp
ITTS_TILE256=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s2mel -o s -- python $GRAFT_REPO_ROOT/tools/s2mel_bench.py 8 800 1926 5 bf16 > $GRAFT_REPO_ROOT/$O/s2mel_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_s2mel -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/s2mel_kernel_stats_tile256.csv
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_codec.py -m gpu -q -s > $O/pytest_pipeline.log 2>&1; echo "pytest_pipeline rc=$?" >> $O/status.txt
cat $O/status.txt; tail -5 $O/pytest_tile.log; tail -3 $O/pytest_pipeline.log; cat $O/prefill_bench.log; cat $O/s2mel_bench.log
