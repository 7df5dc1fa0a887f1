#!/bin/bash
# NEXT ROUND, first GPU call (about one minute): the sampler's phase stamps for the three top-k selections -- radix select (product default in
# sample_kernel), ballot bisection (product default in the beam kernel), and the lower-bound + compaction variant (-DITTS_TOPK_V2, microbench
# build only) -- with the token fingerprint that must agree across them.  Build first (see the header of tools/microbench/sample_stamps.hip):
#   hipcc ... -DITTS_SAMPLE_STAMPS                  tools/microbench/sample_stamps.hip -o tools/microbench/bin/sample_stamps
#   hipcc ... -DITTS_SAMPLE_STAMPS -DITTS_TOPK_V2 -DITTS_SAMPLE_ROWS_V2   tools/microbench/sample_stamps.hip -o tools/microbench/bin/sample_stamps_v2
#   (v2 = lower-bound top-k + row staging with every load in flight)
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04a
mkdir -p $O
for B in 1 64; do
  ITTS_SAMPLE_RADIX=1 timeout 30 tools/microbench/bin/sample_stamps $B 2>&1 | sed "s/^/[radix] /" >> $O/sample_stamps.log
  ITTS_SAMPLE_RADIX=0 timeout 30 tools/microbench/bin/sample_stamps $B 2>&1 | sed "s/^/[bisection] /" >> $O/sample_stamps.log
  ITTS_SAMPLE_RADIX=0 timeout 30 tools/microbench/bin/sample_stamps_v2 $B 2>&1 | sed "s/^/[bisection v2] /" >> $O/sample_stamps.log
done
grep -E "sample_kernel|top-k|fingerprint" $O/sample_stamps.log
# decode attention with four key groups in flight per wave (ITTS_ATTN_UNROLL=4; bitwise test first, then ms/token at 1 and 64 rows)
ITTS_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_gpt.py -q -k unrolled_decode_attention > $O/pytest_attn_unroll.log 2>&1; tail -2 $O/pytest_attn_unroll.log
for B in 1 64; do for u in 1 4 1 4; do
  ITTS_ATTN_UNROLL=$u ITTS_BEAM_BENCH_MODES=1 timeout 200 python tools/beam_bench.py $B 400 2>&1 | grep "^B=" | cut -c1-110 | sed "s/^/unroll=$u /" >> $O/attn_unroll.log
done; done
cat $O/attn_unroll.log
