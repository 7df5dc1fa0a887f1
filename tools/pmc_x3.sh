#!/bin/bash
# GPU box: how busy is the matrix pipe under the fp32x3 kernels, and at what clock?  One rocprofv3 counter pass (SQ_VALU_MFMA_BUSY_CYCLES,
# GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES; --kernel-trace only) over tools/s2mel_bench.py at B utterances x (517 + 1926) frames, two Euler steps, per kernel family:
#   GRBM_GUI_ACTIVE comes back SUMMED over the 8 XCDs (each XCD's GRBM counts its own active cycles), SQ_VALU_MFMA_BUSY_CYCLES summed over all SIMDs:
#   cycles    = GRBM_GUI_ACTIVE / 8                                                 -- the kernel's own clock cycles (per XCD)
#   mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)                    -- fraction of those cycles with the matrix pipe busy
#   clock     = cycles / kernel duration                                            -- the effective engine clock while it ran (DVFS)
# usage: tools/pmc_x3.sh [B=8]  ->  gpurun_out/pmc_x3/x3_pmc.json
set -u
B=${1:-8}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_x3
mkdir -p "$OUT"
rm -f "$OUT/x3_pmc.json"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/raw" -o p -- python $ROOT/tools/s2mel_bench.py $B 517 1926 2 fp32x3 fp32 > "$OUT/run.log" 2>&1
grep -q "finite=True" "$OUT/run.log" || { echo "pmc_x3: workload failed: $(tail -3 "$OUT/run.log")" >&2; exit 1; }
cp "$(find "$OUT/raw" -name '*counter_collection.csv' | head -1)" "$OUT/cc.csv" 2>/dev/null
cp "$(find "$OUT/raw" -name '*kernel_trace.csv' | head -1)" "$OUT/kt.csv" 2>/dev/null
rm -rf "$OUT/raw"
python3 - "$OUT" "$B" <<'PY'
import csv, json, sys, collections
out, B = sys.argv[1], int(sys.argv[2])
FAM = ("gemm_x3", "flash_attn_x3_kernel", "gemm_prefill_kernel", "flash_attn_f32_kernel")      # gemm_x3: gemm_x3w8_kernel (8 waves, the default) and gemm_x3_kernel
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
res = {"B": B, "frames": 517 + 1926, "note": "sums over the launches of two CFG Euler steps per mode (fp32x3 then fp32) inside ONE counter pass: durations under "
       "counter collection are inflated and are used only for the ratio clock = GRBM_GUI_ACTIVE / duration"}
for k in FAM:
    if dur[k][1] == 0: continue
    busy, gui, ns = cnt[k]["SQ_VALU_MFMA_BUSY_CYCLES"], cnt[k]["GRBM_GUI_ACTIVE"], dur[k][0]
    cyc = gui / 8.0                                                  # per-XCD: the counter is summed over the 8 XCDs
    res[k] = {"dispatches": dur[k][1], "duration_ns": ns, "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE_sum_over_8_xcds": gui,
              "cycles_per_xcd": cyc, "mfma_busy_of_own_cycles": busy / (1024.0 * cyc) if cyc else None,
              "effective_clock_GHz": cyc / ns if ns else None, "mfma_busy_at_2p4GHz": busy / (1024.0 * ns * 2.4) if ns else None}
if "gemm_x3" not in res: print("pmc_x3: no gemm_x3 dispatches", file=sys.stderr); sys.exit(1)
json.dump(res, open(f"{out}/x3_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
