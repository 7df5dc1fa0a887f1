#!/bin/bash
# GPU box: HBM traffic of conv_mfma_kernel over ONE BigVGAN forward at the bench shape (B utterances x 1926 frames), from
# PMC counters in two separate passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md section HBM).  FETCH_SIZE is calibrated
# in the same process on aa_act_kernel (known byte count, same coalesced-dword access width): gfx950 under-reports reads.
set -u
B=${1:-64}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_bench
mkdir -p "$OUT"
cat > /tmp/pmc_voc.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from indextts_amd import bigvgan, synth
bh = dict(synth.BIGVGAN_V2_22K)
voc = bigvgan.BigVGAN(bh); voc.load_state_dict(synth.bigvgan_weights(bh)); voc.to("cuda:0")
mel = (torch.randn($B, 80, 1926) * 2 - 4).cuda()
w = voc(mel); torch.cuda.synchronize()
print("done", tuple(w.shape))
PY
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/raw_$ctr" -o p -- python /tmp/pmc_voc.py > "$OUT/run_$ctr.log" 2>&1
  cp "$(find "$OUT/raw_$ctr" -name '*counter_collection.csv' | head -1)" "$OUT/cc_$ctr.csv" 2>/dev/null
  rm -rf "$OUT/raw_$ctr"
done
python3 - "$OUT" "$B" <<'PY'
import csv, json, sys, collections
out, B = sys.argv[1], int(sys.argv[2])
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f"{out}/cc_{ctr}.csv")):
        if r.get("Counter_Name") == ctr:
            k = "conv_mfma" if "conv_mfma" in r["Kernel_Name"] else ("aa_act" if "aa_act" in r["Kernel_Name"] else None)
            if k:
                agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    res[ctr] = {k: {"kb_sum": v[0], "dispatches": v[1]} for k, v in agg.items()}
# calibration on the activation kernel: it reads exactly one f32 tensor per launch; algorithmic bytes over the forward:
T = 1926
act_elems = 0
t, c0 = T, 1536
for i, u in enumerate([4, 4, 2, 2, 2, 2]):
    t *= u; ch = c0 >> (i + 1)
    act_elems += 18 * ch * t
act_elems += (c0 >> 6) * t
act_bytes = act_elems * 4.0 * B
cal = act_bytes / (res["FETCH_SIZE"]["aa_act"]["kb_sum"] * 1024.0)
wcal = act_bytes / (res["WRITE_SIZE"]["aa_act"]["kb_sum"] * 1024.0)
conv_f = res["FETCH_SIZE"]["conv_mfma"]; conv_w = res["WRITE_SIZE"]["conv_mfma"]
n = conv_f["dispatches"]
summary = {"B": B, "mel_frames": T, "conv_dispatches": n,
           "fetch_kb_sum_raw": conv_f["kb_sum"], "write_kb_sum_raw": conv_w["kb_sum"],
           "fetch_calibration_factor": cal, "write_calibration_factor": wcal,
           "hbm_bytes_per_conv_dispatch": (conv_f["kb_sum"] * 1024.0 * cal + conv_w["kb_sum"] * 1024.0 * wcal) / n,
           "note": "per kernel dispatch (a ConvTranspose1d layer is u dispatches); FETCH_SIZE corrected by the factor measured on aa_act_kernel (known bytes, same coalesced dword loads) in the same run"}
json.dump(summary, open(f"{out}/conv_traffic.json", "w"), indent=1)
print(json.dumps(summary))
PY
rm -f "$OUT"/cc_*.csv
