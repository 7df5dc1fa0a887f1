#!/bin/bash
# round 3, call 21: LayerNorm-fused decode GEMMs (1-4 rows): bitwise tests, then decode ms/token at B = 1 (sampled, 3-beam) and B = 4 with the
# fusion off / on, and the configs[0] line
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03u
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gpt.py -x -q -k "fused or layernorm or kernels_agree or goldens or greedy or sample or beam" > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/status.txt
for B in 1 4; do for f in 0 1 0 1; do
  ITTS_DECODE_FUSE_LN=$f ITTS_BEAM_BENCH_MODES=2 timeout 200 python tools/beam_bench.py $B 200 2>&1 | grep "^B=" | cut -c1-120 | sed "s/^/fuse=$f /" >> $O/decode_fuse.log
done; done
for f in 0 1; do ITTS_DECODE_FUSE_LN=$f timeout 300 python tools/config0_check.py 2>&1 | grep "^bf16" | cut -c1-330 | sed "s/^/fuse=$f /" >> $O/decode_fuse.log; done
cat $O/status.txt; tail -4 $O/pytest.log | cut -c1-300; cat $O/decode_fuse.log
