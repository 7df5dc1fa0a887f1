#!/bin/bash
# round 4, GPU call 1: GPT tests on the new attention / fused-LN kernels, decode ms/token by batch and option, one decode-step timeline at 8
# rows, the repaired GPT PMC tool at 8 rows.
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_compaction.py tests/test_gpu_edges.py -x -q > $O/pytest_gpt.log 2>&1; echo "pytest gpt rc=$?" | tee $O/status.txt; tail -5 $O/pytest_gpt.log
timeout 600 python tools/decode_bench.py 560 1,4,8,16,32,64 attn_waves=4 attn_waves=8 attn_waves=16 decode_fuse_ln=0 > $O/decode_bench.log 2>&1; echo "decode_bench rc=$?" | tee -a $O/status.txt
cat $O/decode_bench.log | grep "^B="
timeout 300 bash tools/trace_decode.sh 8 400 > $O/trace8.log 2>&1; cp gpurun_out/trace_decode/step_timeline.txt $O/decode_step_timeline_b8.txt 2>/dev/null; head -14 $O/decode_step_timeline_b8.txt; tail -2 $O/decode_step_timeline_b8.txt
timeout 600 bash tools/pmc_gpt.sh 8 24 > $O/pmc_gpt_b8.log 2>&1; echo "pmc_gpt b8 rc=$?" | tee -a $O/status.txt; cp gpurun_out/pmc_gpt/gpt_pmc_b8.json $O/ 2>/dev/null; tail -30 $O/pmc_gpt_b8.log | head -60
