#!/bin/bash
# round 4, GPU call 9: the wide LayerNorm-fused decode GEMM (5-16 rows: weights on waves 0-3, LayerNorm on waves 4-7, 2 / 4 n-tiles per block) --
# bitwise tests against the two launches it replaces, whole decode loops, then ms / token at 5-16 rows against the separate-LayerNorm path.
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04h
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gpt.py -x -q -k "layernorm_fused or fused_layernorm" > $O/pytest_lnw.log 2>&1; echo "pytest lnw rc=$?" | tee $O/status.txt
tail -4 $O/pytest_lnw.log
timeout 600 python tools/decode_bench.py 560 5,8,12,16 decode_fuse_ln=2 decode_fuse_ln=2,decode_ln_nt=2 decode_fuse_ln=2,decode_ln_nt=0 > $O/decode_bench.log 2>&1; echo "decode_bench rc=$?" | tee -a $O/status.txt
grep "^B=" $O/decode_bench.log
