#!/bin/bash
# round 3, call 26: configs[0] (greedy, B = 1) with the sampler's row prefetch off / on, alternating
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03z
mkdir -p $O
for p in 0 1 0 1; do ITTS_SAMPLE_PREFETCH=$p timeout 200 python tools/config0_check.py 2>&1 | grep "^bf16" | cut -c1-120 | sed "s/^/prefetch=$p /" >> $O/config0_ab.log; done
cat $O/config0_ab.log
