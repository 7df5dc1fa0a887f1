#!/bin/bash
# GPU box: PMC evidence for the f16 x 3 vocoder convs (one BigVGAN forward, B = 8 x 1926 frames):
#   pass A  SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE   -> matrix-pipe busy fraction per kernel
#   pass B  SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE                          -> LDS conflict share (the k-group-major chunks claim 0)
# Each pass is its own rocprofv3 run with --kernel-trace only.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_voc_h3
mkdir -p "$OUT"
cat > /tmp/pmc_voc.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from indextts_amd import bigvgan, synth
bh = dict(synth.BIGVGAN_V2_22K)
voc = bigvgan.BigVGAN(bh, conv_mode="f16x3"); voc.load_state_dict(synth.bigvgan_weights(bh)); voc.to("cuda:0")
mel = (torch.randn(8, 80, 1926, generator=torch.Generator().manual_seed(0)) * 2 - 4).cuda()
voc(mel); torch.cuda.synchronize(); print("done")
PY
cd /tmp && export TMPDIR=/tmp
for pass in "A SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "B SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  set -- $pass; tag=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/raw_$tag" -o p -- python /tmp/pmc_voc.py > "$OUT/run_$tag.log" 2>&1
  cp "$(find "$OUT/raw_$tag" -name '*counter_collection.csv' | head -1)" "$OUT/cc_$tag.csv" 2>/dev/null
  rm -rf "$OUT/raw_$tag"
done
python3 - "$OUT" <<'PY'
import csv, json, sys, collections
out = sys.argv[1]
res = {}
for tag in "AB":
    try:
        rows = list(csv.DictReader(open(f"{out}/cc_{tag}.csv")))
    except Exception as e:
        res[tag] = repr(e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[k]["_n"] += 1
    res[tag] = {k: dict(v) for k, v in agg.items()}
json.dump(res, open(f"{out}/pmc_voc_h3.json", "w"), indent=1)
for tag, d in res.items():
    if isinstance(d, str): print(tag, d); continue
    for k, v in d.items():
        if "conv_h3" in k or "split" in k or "conv_mfma" in k:
            if tag == "A":
                print(k, "mfma_busy/busy_cu=%.3f" % (v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1, v.get("SQ_BUSY_CU_CYCLES", 1))), {a: int(b) for a, b in v.items()})
            else:
                print(k, "lds_conflict/active=%.4f" % (v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, v.get("SQ_LDS_IDX_ACTIVE", 1))), {a: int(b) for a, b in v.items()})
PY
rm -f "$OUT"/cc_*.csv
