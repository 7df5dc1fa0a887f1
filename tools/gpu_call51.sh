#!/bin/bash
# round 3, call 17: the 256 x 256 tile kernel on the f32 MFMA (ITTS_F32_TILE256=1) against the 128 x 128 one: rate + output fingerprints
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03q
mkdir -p $O
for v in 0 1 0 1; do ITTS_F32_TILE256=$v timeout 60 tools/microbench/bin/ga_0 312704 5 2>&1 | sed "s/^/tile256=$v /" >> $O/gemm_f32_tile256.log; done
for M in 39088 4886; do for v in 0 1; do ITTS_F32_TILE256=$v timeout 60 tools/microbench/bin/ga_0 $M 20 2>&1 | sed "s/^/tile256=$v /" >> $O/gemm_f32_tile256.log; done; done
cat $O/gemm_f32_tile256.log | cut -c1-170
