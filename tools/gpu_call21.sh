#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02u
mkdir -p $O
for rep in 1 2; do for b in fa_q2_o5 fa_q2_o7 fa_q1_o5 fa_q1_o7; do timeout 60 tools/microbench/bin/$b 64 2443 | sed "s/^/$b /" >> $O/flash_opt.log 2>&1; done; done
cat $O/flash_opt.log
