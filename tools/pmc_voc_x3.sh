#!/bin/bash
# GPU box: is the bf16 x 3 vocoder conv (conv_x3w_kernel) at the socket's power limit like the x3 GEMM / attention, or stalled below it?  One rocprofv3
# counter pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; --kernel-trace only) over tools/voc_h3_bench.py B bf16x3:96 (BigVGAN forwards at B x 1926
# frames), per kernel family: matrix-pipe busy fraction of the kernel's own cycles and its effective clock (formulas: tools/pmc_x3.sh).
# usage: tools/pmc_voc_x3.sh [B=16]  ->  gpurun_out/pmc_voc_x3/voc_x3_pmc.json
set -u
B=${1:-16}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_voc_x3
mkdir -p "$OUT"
rm -f "$OUT/voc_x3_pmc.json"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/raw" -o p -- python $ROOT/tools/voc_h3_bench.py $B bf16x3:96 > "$OUT/run.log" 2>&1
grep -q "ms / forward" "$OUT/run.log" || { echo "pmc_voc_x3: workload failed: $(tail -3 "$OUT/run.log")" >&2; exit 1; }
cp "$(find "$OUT/raw" -name '*counter_collection.csv' | head -1)" "$OUT/cc.csv" 2>/dev/null
cp "$(find "$OUT/raw" -name '*kernel_trace.csv' | head -1)" "$OUT/kt.csv" 2>/dev/null
rm -rf "$OUT/raw"
python3 - "$OUT" "$B" <<'PY'
import csv, json, sys, collections
out, B = sys.argv[1], int(sys.argv[2])
FAM = ("conv_x3w_kernel<2>", "conv_x3w_kernel<1>", "conv_mfma_kernel", "aa_act_planes_kernel", "aa_act_kernel_v2")
def fam(n):
    for k in FAM:
        if k in n: return k
    return None
dur = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f"{out}/kt.csv")):
    k = fam(r["Kernel_Name"])
    if k: dur[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); dur[k][1] += 1
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f"{out}/cc.csv")):
    k = fam(r["Kernel_Name"])
    if k: cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
res = {"B": B, "frames": 1926, "note": "sums over the launches of the BigVGAN forwards of tools/voc_h3_bench.py inside ONE counter pass; durations under counter "
       "collection are used only for the ratio clock = (GRBM_GUI_ACTIVE / 8 XCDs) / duration"}
for k in FAM:
    if dur[k][1] == 0: continue
    busy, gui, ns = cnt[k]["SQ_VALU_MFMA_BUSY_CYCLES"], cnt[k]["GRBM_GUI_ACTIVE"], dur[k][0]
    cyc = gui / 8.0
    res[k] = {"dispatches": dur[k][1], "duration_ns": ns, "cycles_per_xcd": cyc, "mfma_busy_of_own_cycles": busy / (1024.0 * cyc) if cyc else None,
              "effective_clock_GHz": cyc / ns if ns else None}
json.dump(res, open(f"{out}/voc_x3_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -f "$OUT/cc.csv" "$OUT/kt.csv"
