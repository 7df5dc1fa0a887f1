#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02t
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_s2mel.py -m gpu -q -s > $O/pytest_s2mel.log 2>&1; echo "pytest_s2mel rc=$?" > $O/status.txt
for b in fa_q1_m0 fa_q2_m0; do timeout 60 tools/microbench/bin/$b 64 2443 >> $O/flash.log 2>&1; done
for qs in 1 2; do
ITTS_FA_QS=$qs timeout 300 python tools/s2mel_bench.py 8 800 1926 25 bf16 2>&1 | grep "ms total" | sed "s/^/QS=$qs /" >> $O/s2mel_bench.log
ITTS_FA_QS=$qs timeout 300 python tools/s2mel_bench.py 32 517 1926 25 bf16 2>&1 | grep "ms total" | sed "s/^/QS=$qs /" >> $O/s2mel_bench.log
done
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_fa -o p -- $GRAFT_REPO_ROOT/tools/microbench/bin/fa_q2_m0 64 2443 > $GRAFT_REPO_ROOT/$O/pmc_run.log 2>&1
python3 - /tmp/pmc_fa >> $GRAFT_REPO_ROOT/$O/pmc_summary.txt <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(float); n=set()
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'flash' in r['Kernel_Name']:
            d[r['Counter_Name']]+=float(r['Counter_Value']); n.add(r['Dispatch_Id'])
k=max(len(n),1)
print({a:round(b/k) for a,b in d.items()})
PY
cd $GRAFT_REPO_ROOT
cat $O/status.txt; tail -2 $O/pytest_s2mel.log; cat $O/flash.log $O/s2mel_bench.log $O/pmc_summary.txt
