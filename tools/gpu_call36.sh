#!/bin/bash
# round 3, call 1: grid-barrier microbenchmark (decides the decode redesign) + what the CURRENT f32 s2mel mode costs at the bench's
# frame count (small batch: its attention is one wave per query), next to bf16 on the same shape.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03a
mkdir -p $O
timeout 60 tools/microbench/bin/grid_barrier 4000 > $O/grid_barrier.log 2>&1; echo "grid_barrier rc=$?" > $O/status.txt
timeout 120 python tools/s2mel_bench.py 2 517 1926 1 fp32 > $O/s2mel_f32_b2.log 2>&1; echo "s2mel f32 rc=$?" >> $O/status.txt
timeout 60 python tools/s2mel_bench.py 2 517 1926 1 bf16 > $O/s2mel_bf16_b2.log 2>&1; echo "s2mel bf16 rc=$?" >> $O/status.txt
cat $O/status.txt; cat $O/grid_barrier.log; tail -2 $O/s2mel_f32_b2.log; tail -2 $O/s2mel_bf16_b2.log
