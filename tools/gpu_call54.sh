#!/bin/bash
# round 3, call 20: the configs[0] bench extra alone
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03t
mkdir -p $O
timeout 300 python tools/config0_check.py > $O/config0.log 2>&1; echo "rc=$?" >> $O/config0.log
tail -12 $O/config0.log | cut -c1-700
