#!/bin/bash
# round 3, call 10: f32 tile GEMM with the weight fragments in registers (bitwise A/B + timing), the IndexTTS-2 class tests, x3 regression.
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03l
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_pipeline.py tests/test_gpu_gemm_x3.py tests/test_gpu_gpt.py -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/status.txt
timeout 120 python tools/gemm_x3_bench.py 312704 3 > $O/gemm_bench_wreg.log 2>&1
ITTS_F32_WREG=0 timeout 120 python tools/gemm_x3_bench.py 312704 3 2>&1 | grep " f32:" | sed 's/^/LDS-staged weights: /' >> $O/gemm_bench_wreg.log
timeout 200 python tools/s2mel_bench.py 64 517 1926 3 fp32 2>&1 | tail -1 >> $O/s2mel.log
ITTS_F32_WREG=0 timeout 200 python tools/s2mel_bench.py 64 517 1926 3 fp32 2>&1 | tail -1 | sed 's/^/LDS-staged weights: /' >> $O/s2mel.log
cat $O/status.txt; grep -E "passed|failed|rror|latent pass|f32 s2mel" $O/pytest.log | tail -12 | cut -c1-300; cat $O/gemm_bench_wreg.log | grep -v amdgpu; cat $O/s2mel.log
