#!/bin/bash
# GPU box: PMC evidence for the GPT half of the hot path at the bench shape (64 utterances x 128 text tokens):
#   pass A  SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE  -> MFMA busy fraction of the prefill GEMM kernels
#   pass B  FETCH_SIZE, pass C  WRITE_SIZE              -> HBM bytes of the decode-step kernels (attention, decode GEMMs)
# Each pass is its own rocprofv3 run with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots / HBM).
set -u
NEW=${1:-24}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_gpt
mkdir -p "$OUT"
cat > /tmp/pmc_gpt.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from indextts_amd import gpt, synth
gcfg = dict(synth.GPT_V25)
m = gpt.UnifiedVoice(**gcfg, precision="bf16", device="cuda:0")
m.load_state_dict(synth.gpt_weights(gcfg, suppress_eos=True))
B = 64
g = torch.Generator().manual_seed(0)
text = torch.randint(2, 12000, (B, 128), generator=g).cuda(); langs = torch.full((B,), 3, dtype=torch.long).cuda()
style = torch.randn(1, 192, generator=g).cuda(); emo = (torch.randn(1, 1280, generator=g) * 0.1).cuda()
m.use_graph = False            # counters are collected per dispatch; graph replays are not attributed per kernel
codes, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=$NEW,
                              do_sample=True, top_p=0.8, top_k=30, temperature=0.8, num_beams=1, repetition_penalty=10.0)
torch.cuda.synchronize()
print("done", tuple(codes.shape), m.last_timing)
PY
cd /tmp && export TMPDIR=/tmp
for pass in "A SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "B FETCH_SIZE" "C WRITE_SIZE"; do
  set -- $pass; tag=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/raw_$tag" -o p -- python /tmp/pmc_gpt.py > "$OUT/run_$tag.log" 2>&1
  cp "$(find "$OUT/raw_$tag" -name '*counter_collection.csv' | head -1)" "$OUT/cc_$tag.csv" 2>/dev/null
  cp "$(find "$OUT/raw_$tag" -name '*kernel_trace.csv' | head -1)" "$OUT/kt_$tag.csv" 2>/dev/null
  rm -rf "$OUT/raw_$tag"
done
python3 - "$OUT" "$NEW" <<'PY'
import csv, json, sys, collections
out, new = sys.argv[1], int(sys.argv[2])
def short(n):
    for k in ("gemm_prefill_kernel", "gemm_decode64_kernel", "attn_kernel", "ln_kernel", "sample_kernel", "gemm_kernel"):
        if k in n: return k
    return None
def durations(tag):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f"{out}/kt_{tag}.csv")):
        k = short(r["Kernel_Name"])
        if k: d[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); d[k][1] += 1
    return d
def counters(tag):
    c = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f"{out}/cc_{tag}.csv")):
        k = short(r["Kernel_Name"])
        if k: c[k][r["Counter_Name"]] += float(r["Counter_Value"])
    return c
res = {"new_tokens": new, "note": "sums over one generate call (prefill + new_tokens-1 decode steps, eager launches); durations from the kernel trace of the same pass"}
dA, cA = durations("A"), counters("A")
for k in ("gemm_prefill_kernel",):
    busy, ns, n = cA[k]["SQ_VALU_MFMA_BUSY_CYCLES"], dA[k][0], dA[k][1]
    res[k] = {"dispatches": n, "duration_ns": ns, "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": cA[k]["GRBM_GUI_ACTIVE"],
              "mfma_busy_frac_at_2p4GHz": busy / (1024.0 * ns * 2.4) if ns else None,
              "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration_ns x 2.4 cycles/ns)"}
dB, cB, dC, cC = durations("B"), counters("B"), durations("C"), counters("C")
for k in ("attn_kernel", "gemm_decode64_kernel", "ln_kernel"):
    f_kb, w_kb = cB[k]["FETCH_SIZE"], cC[k]["WRITE_SIZE"]
    ns = dB[k][0]
    res[k] = {"dispatches": dB[k][1], "duration_ns_fetch_pass": ns, "FETCH_SIZE_kb_raw": f_kb, "WRITE_SIZE_kb": w_kb,
              "hbm_GBps_raw": (f_kb + w_kb) * 1024.0 / ns if ns else None,
              "hbm_GBps_fetch_x2": (2 * f_kb + w_kb) * 1024.0 / ns if ns else None,
              "correction": "gfx950 FETCH_SIZE reports half of a wide (16 B/lane) coalesced read stream (guide, HBM section): x2 column"}
json.dump(res, open(f"{out}/gpt_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -f "$OUT"/cc_*.csv "$OUT"/kt_*.csv
