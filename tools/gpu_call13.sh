#!/bin/bash
# GPU box: flash attention with the swizzled LDS layout (variants), QS=2 default; GPT kernels with / without SLP vectorisation
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02m
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_s2mel.py -m gpu -q -s > $O/pytest_s2mel.log 2>&1; echo "pytest_s2mel rc=$?" > $O/status.txt
for b in fa_q1_o0 fa_q1_o5 fa_q2_o0 fa_q2_o1 fa_q2_o4 fa_q2_o5; do timeout 60 tools/microbench/bin/$b 64 2443 | sed "s/^/$b /" >> $O/flash_opt.log 2>&1; done
timeout 300 python tools/s2mel_bench.py 8 800 1926 25 bf16 2>&1 | grep "ms total" >> $O/s2mel_bench.log
ITTS_TILE256=1 timeout 300 python tools/s2mel_bench.py 32 517 1926 25 bf16 2>&1 | grep "ms total" | sed "s/^/TILE256=1 /" >> $O/s2mel_bench.log
ITTS_BEAM_BENCH_MODES=2 timeout 300 python tools/beam_bench.py 64 200 > $O/beam_default.log 2>&1
cp indextts_amd/csrc/libindextts_hip.so /tmp/lib_default.so
cp tools/microbench/bin/libindextts_hip_gptnoslp.so indextts_amd/csrc/libindextts_hip.so
ITTS_BEAM_BENCH_MODES=2 timeout 300 python tools/beam_bench.py 64 200 > $O/beam_gptnoslp.log 2>&1
timeout 200 python tools/prefill_bench.py 2>&1 | grep TFLOP | sed "s/^/gptnoslp /" >> $O/prefill_bench.log
cp /tmp/lib_default.so indextts_amd/csrc/libindextts_hip.so
timeout 200 python tools/prefill_bench.py 2>&1 | grep TFLOP | sed "s/^/default /" >> $O/prefill_bench.log
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_fa_$i -o p -- $GRAFT_REPO_ROOT/tools/microbench/bin/fa_q2_o0 64 2443 > $GRAFT_REPO_ROOT/$O/pmc_q2_$i.log 2>&1
  f=$(find /tmp/pmc_fa_$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/pmc_q2_$i.csv
done
cd $GRAFT_REPO_ROOT
cat $O/status.txt; tail -2 $O/pytest_s2mel.log; cat $O/flash_opt.log $O/s2mel_bench.log; tail -4 $O/beam_default.log; tail -4 $O/beam_gptnoslp.log; cat $O/prefill_bench.log
