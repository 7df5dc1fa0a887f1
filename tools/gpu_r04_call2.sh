#!/bin/bash
# round 4, GPU call 2: weight-prefetch microbenchmark (does touching the next kernel's weights shorten the LN -> GEMM chain?), the 6-product
# x3 GEMM against an f64 product on every s2mel shape, the three s2mel modes against the reference classes' production-width fixture.
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04b
mkdir -p $O
timeout 150 tools/microbench/bin/weight_prefetch > $O/weight_prefetch.log 2>&1; echo "weight_prefetch rc=$?" | tee $O/status.txt
cat $O/weight_prefetch.log
timeout 600 python -m pytest tests/test_gpu_gemm_x3.py -x -q -s > $O/pytest_x3.log 2>&1; echo "pytest x3 rc=$?" | tee -a $O/status.txt
grep -E "GEMM|variant|passed|failed|Error" $O/pytest_x3.log | tail -30
timeout 600 python -m pytest tests/test_gpu_s2mel.py -x -q -s -k "production" > $O/pytest_s2mel_prod.log 2>&1; echo "pytest s2mel prod rc=$?" | tee -a $O/status.txt
grep -E "production|passed|failed|Error" $O/pytest_s2mel_prod.log | tail -12
