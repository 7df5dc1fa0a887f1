#!/bin/bash
# round 4, GPU call 14: 32-row decode GEMM blocks (80 KiB slab, two blocks per CU) against the 64-row form above 32 rows
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04n
mkdir -p $O
timeout 600 python tools/decode_bench.py 560 40,48,64 decode_mt=2 decode_mt=2,decode_nt=0 > $O/decode_bench.log 2>&1; echo "decode_bench rc=$?" | tee $O/status.txt
grep "^B=" $O/decode_bench.log
