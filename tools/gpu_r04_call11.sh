#!/bin/bash
# round 4, GPU call 11: which stage of the bf16 s2mel estimator is not bit-stable?  (engine trace checksums, 24 repetitions per setting)
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04j
mkdir -p $O
timeout 600 python tools/s2mel_trace.py 2 517 1926 24 bf16 bf16:s2mel_fused=0 bf16:tile256=0 fp32x3 > $O/trace_b2.log 2>&1; echo "trace rc=$?" | tee $O/status.txt
grep -v "amdgpu.ids" $O/trace_b2.log | cut -c1-400
