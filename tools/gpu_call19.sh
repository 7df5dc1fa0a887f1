#!/bin/bash
# GPU box: which LDS access of the flash kernel conflicts (K reads / V^T reads / staging writes removed one at a time)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02s
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 0 64 128 256 192 448; do
  timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc_fa_$m -o p -- $GRAFT_REPO_ROOT/tools/microbench/bin/fa_q2_m$m 64 2443 > $GRAFT_REPO_ROOT/$O/run_$m.log 2>&1
  python3 - /tmp/pmc_fa_$m $m >> $GRAFT_REPO_ROOT/$O/summary.txt <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(float); n=set()
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'flash' in r['Kernel_Name']:
            d[r['Counter_Name']]+=float(r['Counter_Value']); n.add(r['Dispatch_Id'])
k=max(len(n),1)
print('FA_ABL',sys.argv[2], {a:round(b/k) for a,b in d.items()})
PY
done
cat $GRAFT_REPO_ROOT/$O/summary.txt
