#!/bin/bash
# round 3, call 8: where does the f32x3 GEMM's time go?  GEMM-only timings + two PMC passes (SQ counters) on one shape.
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r03h
mkdir -p $O
timeout 120 python tools/gemm_x3_bench.py 39088 5 > $O/gemm_bench.log 2>&1
ITTS_X3_PRODUCTS=6 timeout 120 python tools/gemm_x3_bench.py 39088 5 2>&1 | grep f32x3 | sed 's/^/6-product: /' >> $O/gemm_bench.log
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_x3_$i -o p -- python $GRAFT_REPO_ROOT/tools/gemm_x3_bench.py 39088 1 > $O/pmc_run_$i.log 2>&1
  f=$(find /tmp/pmc_x3_$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && grep -E "Counter_Name|gemm_x3|gemm_prefill" "$f" > $O/pmc_$i.csv
done
cat $O/gemm_bench.log
python3 - $O <<'PY'
import csv, sys, collections
o = sys.argv[1]
for i in (1, 2):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    try:
        for r in csv.DictReader(open(f"{o}/pmc_{i}.csv")):
            k = ("x3" if "gemm_x3" in r["Kernel_Name"] else "f32") + " grid" + r["Grid_Size"]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    except Exception as e:
        print("pass", i, "unreadable", e); continue
    for k, d in sorted(acc.items()):
        print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
