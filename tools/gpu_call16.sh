#!/bin/bash
# GPU box: PMC counters of the vocoder conv kernel and the tile GEMMs (where do the non-MFMA cycles go)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_voc_$i -o p -- python $GRAFT_REPO_ROOT/tools/voc_stage_profile.py 8 > $GRAFT_REPO_ROOT/$O/pmc_voc_$i.log 2>&1
  python3 - /tmp/pmc_voc_$i $GRAFT_REPO_ROOT/$O/pmc_voc_$i.json <<'PY'
import csv,glob,json,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:60]
        d[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
json.dump({k:dict(v, dispatches=len(n[k])) for k,v in d.items()}, open(sys.argv[2],'w'), indent=1)
PY
  ITTS_TILE256=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_s2_$i -o p -- python $GRAFT_REPO_ROOT/tools/s2mel_bench.py 16 517 1926 2 bf16 > $GRAFT_REPO_ROOT/$O/pmc_s2_$i.log 2>&1
  python3 - /tmp/pmc_s2_$i $GRAFT_REPO_ROOT/$O/pmc_s2_$i.json <<'PY'
import csv,glob,json,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:60]
        d[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
json.dump({k:dict(v, dispatches=len(n[k])) for k,v in d.items()}, open(sys.argv[2],'w'), indent=1)
PY
done
ls $GRAFT_REPO_ROOT/$O
