#!/bin/bash
# round 3, call 16: register-staged variant of the f32 tile GEMM (microbench bit 512: global_load -> ds_write instead of LDS-DMA)
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03p
mkdir -p $O
for m in 0 512 520 0 512; do timeout 60 tools/microbench/bin/ga_$m 312704 5 >> $O/gemm_f32_regstage.log 2>&1; done
cat $O/gemm_f32_regstage.log | cut -c1-160
