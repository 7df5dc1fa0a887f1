#!/bin/bash
# round 4, GPU call 3: the fp32x3 mode's attention on bf16 planes (flash_attn_x3_kernel): unit tests against an f64 attention and the native f32
# flash kernel, the s2mel suite (every mode against the reference classes' fixtures), one solve at the bench's per-utterance shape per mode.
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04c
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_attn_x3.py -x -q -s > $O/pytest_attn_x3.log 2>&1; echo "pytest attn_x3 rc=$?" | tee $O/status.txt
grep -E "attention|passed|failed|Error|error" $O/pytest_attn_x3.log | tail -12
timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_gemm_x3.py -x -q -s > $O/pytest_s2mel.log 2>&1; echo "pytest s2mel rc=$?" | tee -a $O/status.txt
grep -E "production|passed|failed|Error" $O/pytest_s2mel.log | tail -12
timeout 600 python tools/s2mel_bench.py 8 517 1926 5 fp32 fp32x3:x3_attn=0,x3_products=8 fp32x3:x3_attn=0,x3_products=6 fp32x3:x3_products=8 fp32x3:x3_products=6 > $O/s2mel_bench.log 2>&1; echo "s2mel_bench rc=$?" | tee -a $O/status.txt
grep "^B=" $O/s2mel_bench.log
