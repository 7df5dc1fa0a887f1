#!/bin/bash
# round 3, call 29: sample_kernel phase stamps with the ballot-bisection top-k (ITTS_SAMPLE_RADIX=0)
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03za
mkdir -p $O
for B in 1 64; do ITTS_SAMPLE_RADIX=0 timeout 30 tools/microbench/bin/sample_stamps $B 2>&1 | sed "s/^/[bisection] /" >> $O/sample_stamps.log; done
tail -20 $O/sample_stamps.log
