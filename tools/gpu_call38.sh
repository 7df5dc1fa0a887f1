#!/bin/bash
# round 3, call 3: the f32 fast path (f32-MFMA tile GEMMs with fused epilogues + f32 flash attention) and the full-size parity tests,
# then the first bench line with the fp32-CFM headline beside the bf16 one.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03c
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_fullsize.py -x -q -s > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" > $O/status.txt
timeout 600 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_pipeline.py tests/test_gpu_cond.py -x -q > $O/pytest_gpt.log 2>&1; echo "pytest gpt/pipeline rc=$?" >> $O/status.txt
timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/status.txt
cat $O/status.txt; grep -E "rms|max\|d\||passed|failed|error|Error" $O/pytest_new.log | tail -30; tail -3 $O/pytest_gpt.log; tail -5 $O/bench.err; cut -c1-600 $O/bench.json
