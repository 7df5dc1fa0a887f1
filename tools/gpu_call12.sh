#!/bin/bash
# GPU box: flash attention variants (FA_OPT bits) + PMC counters of the attention kernel; vocoder with / without SLP vectorisation
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02l
mkdir -p $O
for qs in 1 2; do for o in 0 1 2 4 7; do timeout 60 tools/microbench/bin/fa_q${qs}_o${o} 64 2443 | sed "s/^/FA_OPT=$o /" >> $O/flash_opt.log 2>&1; done; done
timeout 300 python tools/voc_stage_profile.py 16 > $O/voc_default.log 2>&1
cp indextts_amd/csrc/libindextts_hip.so /tmp/lib_default.so
cp tools/microbench/bin/libindextts_hip_vocnoslp.so indextts_amd/csrc/libindextts_hip.so
timeout 300 python tools/voc_stage_profile.py 16 > $O/voc_noslp.log 2>&1
cp /tmp/lib_default.so indextts_amd/csrc/libindextts_hip.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_list.txt 2>&1
for qs in 1 2; do
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA" \
             "SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_fa_${qs}_$i -o p -- $GRAFT_REPO_ROOT/tools/microbench/bin/fa_q${qs}_o0 64 2443 > $GRAFT_REPO_ROOT/$O/pmc_q${qs}_$i.log 2>&1
    f=$(find /tmp/pmc_fa_${qs}_$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/pmc_q${qs}_$i.csv
  done
done
cd $GRAFT_REPO_ROOT
cat $O/flash_opt.log; head -3 $O/voc_default.log; grep -i "act" $O/voc_default.log | head -8; head -3 $O/voc_noslp.log; grep -i "act" $O/voc_noslp.log | head -8; ls $O
