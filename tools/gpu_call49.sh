#!/bin/bash
# round 3, call 15: loader-wave variant of the f32 tile GEMM (microbench bit 256) against the baseline
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03o
mkdir -p $O
for m in 0 256 264 0 256; do timeout 60 tools/microbench/bin/ga_$m 312704 5 >> $O/gemm_f32_loader.log 2>&1; done
cat $O/gemm_f32_loader.log | cut -c1-160
