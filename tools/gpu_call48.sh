#!/bin/bash
# round 3, call 14 (experiments): f32 tile-GEMM ablation microbench (what limits the main loop; start stagger; interleaved K-tile schedule),
# the same switches on the real f32 solve (bit fingerprints must agree), decode slab-DMA rotation A/B, per-kernel timeline of a decode step.
set -u
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/r03n
mkdir -p $O
for m in 0 1 2 4 8 16 64 128 129 136 144 192; do
  timeout 60 tools/microbench/bin/ga_$m 312704 5 >> $O/gemm_f32_ablate.log 2>&1
done
run_s2() { env "$@" timeout 200 python tools/s2mel_bench.py 64 517 1926 3 fp32 2>&1 | tail -1 | sed "s/^/[$*] /" >> $O/s2mel_ab.log; }
run_s2 ITTS_NOP=1
run_s2 ITTS_F32_ILV=1
run_s2 ITTS_F32_STAGGER=1
run_s2 ITTS_F32_STAGGER=2
run_s2 ITTS_FA32_ILV=1
run_s2 ITTS_F32_ILV=1 ITTS_F32_STAGGER=2 ITTS_FA32_ILV=1
for rot in 0 1 0 1; do
  ITTS_DECODE_ROT=$rot ITTS_BEAM_BENCH_MODES=1 timeout 200 python tools/beam_bench.py 64 400 2>&1 | tail -1 | sed "s/^/rot=$rot /" >> $O/decode_rot.log
done
timeout 300 bash tools/trace_decode.sh 64 > $O/trace_decode.log 2>&1
cp gpurun_out/trace_decode/step_timeline.txt $O/decode_step_timeline_b64.txt 2>/dev/null
cat $O/gemm_f32_ablate.log | cut -c1-160; cat $O/s2mel_ab.log | cut -c1-330; cat $O/decode_rot.log | cut -c1-200; tail -3 $O/trace_decode.log | cut -c1-200
