#!/bin/bash
# round 3, call 23: per-kernel timeline of a decode step at B = 1 (after the LayerNorm fusion)
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03w
mkdir -p $O
timeout 300 bash tools/trace_decode.sh 1 > $O/trace_decode_b1.log 2>&1
cp gpurun_out/trace_decode/step_timeline.txt $O/decode_step_timeline_b1.txt 2>/dev/null
head -12 $O/decode_step_timeline_b1.txt | cut -c1-150; tail -8 $O/decode_step_timeline_b1.txt | cut -c1-150
