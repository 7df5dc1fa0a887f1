#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04k
mkdir -p $O
timeout 600 python tools/s2mel_trace.py 2 517 1926 24 bf16 bf16:dbg=1 bf16:dbg=2 bf16:dbg=4 bf16:dbg=8 > $O/trace_dbg.log 2>&1; echo "trace rc=$?" | tee $O/status.txt
grep -v "amdgpu.ids" $O/trace_dbg.log | grep -v "repetition" | cut -c1-600
DEPTH=1 WN_LAYERS=1 timeout 300 python tools/s2mel_trace.py 2 517 1926 200 bf16 > $O/trace_depth1.log 2>&1; echo "trace depth1 rc=$?" | tee -a $O/status.txt
grep -v "amdgpu.ids" $O/trace_depth1.log | grep -v "repetition" | cut -c1-600
