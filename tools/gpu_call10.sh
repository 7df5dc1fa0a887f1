#!/bin/bash
# GPU box: flash attention with scalar-f32 softmax (no v_pk_*), ablation again
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02k
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_s2mel.py -m gpu -q -s > $O/pytest_s2mel.log 2>&1; echo "pytest_s2mel rc=$?" > $O/status.txt
for qs in 1 2; do for m in 0 1 48 63; do timeout 60 tools/microbench/bin/fa_q${qs}_m${m} 64 2443 >> $O/flash_ablate.log 2>&1; done; done
for qs in 1 2; do
  ITTS_FA_QS=$qs timeout 300 python tools/s2mel_bench.py 8 800 1926 25 bf16 2>&1 | grep "ms total" | sed "s/^/QS=$qs /" >> $O/s2mel_bench.log
  ITTS_FA_QS=$qs ITTS_TILE256=1 timeout 300 python tools/s2mel_bench.py 32 517 1926 25 bf16 2>&1 | grep "ms total" | sed "s/^/TILE256=1 QS=$qs /" >> $O/s2mel_bench.log
done
cat $O/status.txt; tail -3 $O/pytest_s2mel.log; cat $O/flash_ablate.log $O/s2mel_bench.log
