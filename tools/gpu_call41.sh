#!/bin/bash
# round 3, call 6: f32x3 GEMM with the weight fragments loaded straight into registers (no LDS-DMA pieces for W) and the operand split
# software-pipelined under the MFMAs: correctness, then A/B of the interleave and of 6 vs 8 products.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03g
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm_x3.py -x -q -s > $O/pytest.log 2>&1; echo "pytest x3 rc=$?" > $O/status.txt
timeout 200 python tools/s2mel_bench.py 8 517 1926 3 fp32x3 2>&1 | tail -1 | sed 's/^/sched, 8 products: /' >> $O/s2mel_steps.log
ITTS_X3_SCHED=0 timeout 200 python tools/s2mel_bench.py 8 517 1926 3 fp32x3 2>&1 | tail -1 | sed 's/^/no sched, 8 products: /' >> $O/s2mel_steps.log
ITTS_X3_PRODUCTS=6 timeout 200 python tools/s2mel_bench.py 8 517 1926 3 fp32x3 2>&1 | tail -1 | sed 's/^/sched, 6 products: /' >> $O/s2mel_steps.log
timeout 200 python tools/s2mel_bench.py 64 517 1926 2 fp32x3 2>&1 | tail -1 | sed 's/^/B=64 sched, 8 products: /' >> $O/s2mel_steps.log
cat $O/status.txt; grep -E "max-rel|passed|failed|rror|max\|d\|" $O/pytest.log | tail -12 | cut -c1-300; cat $O/s2mel_steps.log
