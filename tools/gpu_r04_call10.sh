#!/bin/bash
# round 4, GPU call 10: the MFMA -> inline-asm hazard microbenchmark, then -- with fa_max3 as compiler-generated code -- the attention / s2mel suites,
# the fused-LayerNorm decode tests at the new default (1-8 rows), and run-to-run determinism of one estimator call per mode (16 repetitions).
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04i
mkdir -p $O
timeout 120 tools/microbench/bin/mfma_asm_hazard > $O/mfma_asm_hazard.log 2>&1; echo "mfma_asm_hazard rc=$?" | tee $O/status.txt
cat $O/mfma_asm_hazard.log
SOLVE=0 timeout 600 python tools/s2mel_determinism.py 2 517 1926 1 16 bf16 fp32x3 fp32 bf16:tile256=0 > $O/determinism_b2.log 2>&1; echo "determinism rc=$?" | tee -a $O/status.txt
grep -E "^poison|first bad" $O/determinism_b2.log | cut -c1-900
timeout 900 python -m pytest tests/test_gpu_s2mel.py tests/test_gpu_attn_x3.py tests/test_gpu_gemm_x3.py -x -q > $O/pytest_s2mel.log 2>&1; echo "pytest s2mel rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_s2mel.log
timeout 900 python -m pytest tests/test_gpu_gpt.py -x -q -k "layernorm_fused or fused_layernorm or invarian or compaction" > $O/pytest_gpt_ln.log 2>&1; echo "pytest gpt ln rc=$?" | tee -a $O/status.txt
tail -3 $O/pytest_gpt_ln.log
