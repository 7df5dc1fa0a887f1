#!/bin/bash
# GPU box: PMC sweep over every kernel of one bench step (LDS conflicts, unit utilisation)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_bench_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-extras --gen-tokens 120 > $GRAFT_REPO_ROOT/$O/pmc_bench_$i.log 2>&1
  python3 - /tmp/pmc_bench_$i $GRAFT_REPO_ROOT/$O/pmc_bench_$i.json <<'PY'
import csv,glob,json,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:70]
        d[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
json.dump({k:dict(v, dispatches=len(n[k])) for k,v in d.items()}, open(sys.argv[2],'w'), indent=1)
PY
done
ls -la $GRAFT_REPO_ROOT/$O; tail -3 $GRAFT_REPO_ROOT/$O/pmc_bench_1.log | cut -c1-300
