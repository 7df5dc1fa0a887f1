#!/bin/bash
# round 3, call 28: phase time stamps of sample_kernel at 1 and 64 rows
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03za
mkdir -p $O
for B in 1 64; do timeout 30 tools/microbench/bin/sample_stamps $B >> $O/sample_stamps.log 2>&1; done
cat $O/sample_stamps.log
