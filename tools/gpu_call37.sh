#!/bin/bash
# round 3, call 2: per-Euler-step cost of the CURRENT f32 s2mel mode at the bench's frame count (the 1-step call of call 1 was dominated by
# the per-solve host setup): B = 8, 1 and 3 steps, f32 and bf16.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03b
mkdir -p $O
for prec in fp32 bf16; do
  for steps in 1 3; do
    timeout 200 python tools/s2mel_bench.py 8 517 1926 $steps $prec 2>&1 | tail -1 >> $O/s2mel_steps.log
  done
done
cat $O/s2mel_steps.log
