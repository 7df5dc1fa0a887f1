#!/bin/bash
# GPU box: PMC evidence for the GPT half of the hot path (north_star: "achieved HBM GB/s for the decode step, MFMA utilisation for the prefill
# GEMMs"), at B utterances x 128 text tokens of the full-size model:
#   pass T  kernel trace only                            -> per-kernel durations (counter collection inflates them: never used for rates)
#   pass A  SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE   -> MFMA-busy fraction of the prefill GEMM kernels
#   pass B  FETCH_SIZE, pass C  WRITE_SIZE               -> HBM bytes per kernel of the decode step (attention, decode GEMMs, LayerNorm)
# Each counter pass is its own rocprofv3 run with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots / HBM).
# FAILS (exit 1, no result file) when the workload of any pass does not finish or a kernel it reports on has no dispatches: round 3's
# version wrote a file of zeros after its workload had crashed at model construction.
# usage: tools/pmc_gpt.sh [B=64] [new_tokens=24]   ->  gpurun_out/pmc_gpt/gpt_pmc_b<B>.json
set -u
B=${1:-64}
NEW=${2:-24}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_gpt
mkdir -p "$OUT"
rm -f "$OUT/gpt_pmc_b$B.json"
cat > /tmp/pmc_gpt.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from indextts_amd import gpt, synth
gcfg = dict(synth.GPT_V25)
m = gpt.UnifiedVoice(**gcfg, spk_cond_mode="campplus", precision="bf16", device="cuda:0")      # as bench.py::HipEngine builds it
m.load_state_dict(synth.gpt_weights(gcfg, seed=1234, suppress_eos=True))
m.post_init_gpt2_config(kv_cache=True, half=True)
B = $B
g = torch.Generator().manual_seed(0)
text = torch.randint(2, 12000, (B, 128), generator=g).cuda(); langs = torch.full((B,), 3, dtype=torch.long).cuda()
style = torch.randn(1, 192, generator=g).cuda(); emo = (torch.randn(1, 1280, generator=g) * 0.1).cuda()
m.use_graph = False            # counters are collected per dispatch; graph replays are not attributed per kernel
codes, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=$NEW,
                              do_sample=True, top_p=0.8, top_k=30, temperature=0.8, num_beams=1, repetition_penalty=10.0)
torch.cuda.synchronize()
assert tuple(codes.shape) == (B, $NEW), codes.shape
print("WORKLOAD_DONE", tuple(codes.shape), m.last_timing)
PY
cd /tmp && export TMPDIR=/tmp
fail=0
for pass in "T" "A SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "B FETCH_SIZE" "C WRITE_SIZE"; do
  set -- $pass; tag=$1; shift
  if [ $# -gt 0 ]; then pmc="--pmc $*"; else pmc=""; fi
  timeout 600 rocprofv3 $pmc --kernel-trace --output-format csv -d "$OUT/raw_$tag" -o p -- python /tmp/pmc_gpt.py > "$OUT/run_b${B}_$tag.log" 2>&1
  rc=$?
  if [ $rc -ne 0 ] || ! grep -q WORKLOAD_DONE "$OUT/run_b${B}_$tag.log"; then
    echo "pmc_gpt: pass $tag FAILED (rc=$rc): $(tail -3 "$OUT/run_b${B}_$tag.log")" >&2; fail=1
  fi
  cp "$(find "$OUT/raw_$tag" -name '*counter_collection.csv' | head -1)" "$OUT/cc_$tag.csv" 2>/dev/null
  cp "$(find "$OUT/raw_$tag" -name '*kernel_trace.csv' | head -1)" "$OUT/kt_$tag.csv" 2>/dev/null
  rm -rf "$OUT/raw_$tag"
done
if [ $fail -ne 0 ]; then rm -f "$OUT"/cc_*.csv "$OUT"/kt_*.csv; echo "pmc_gpt: no result written" >&2; exit 1; fi
python3 - "$OUT" "$B" "$NEW" <<'PY'
import csv, json, sys, collections
out, B, new = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
KERNELS = ("gemm_prefill_kernel", "gemm_tile256_kernel", "gemm_decode64_kernel", "gemm_decode_ln_kernel", "attn_kernel", "ln_kernel", "sample_kernel")
def short(n):
    for k in KERNELS:
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
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(f"{out}/cc_{tag}.csv")):
        k = short(r["Kernel_Name"])
        if k: c[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k] += 1
    return c, n
res = {"B": B, "new_tokens": new,
       "note": "sums over one generate call (prefill + new_tokens-1 decode steps, eager launches); durations from the counter-free kernel-trace pass T; "
               "FETCH_SIZE doubled (gfx950 reports half of a wide 16 B / lane coalesced read stream, MI355X_MICROARCH.md HBM section), WRITE_SIZE as read"}
dT = durations("T")
cA, nA = counters("A")
dA = durations("A")
problems = []
for k in ("gemm_prefill_kernel", "gemm_tile256_kernel"):
    if nA[k] == 0: continue
    busy, ns_a, n = cA[k]["SQ_VALU_MFMA_BUSY_CYCLES"], dA[k][0], dA[k][1]
    gui = cA[k]["GRBM_GUI_ACTIVE"]
    res[k] = {"dispatches": n, "duration_ns_trace_pass": dT[k][0], "duration_ns_counter_pass": ns_a, "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui,
              "mfma_busy_frac_at_2p4GHz": busy / (1024.0 * ns_a * 2.4) if ns_a else None,
              "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration_ns x 2.4 cycles/ns), counter and duration of the SAME (counter) pass"}
if not any(k in res for k in ("gemm_prefill_kernel", "gemm_tile256_kernel")): problems.append("no prefill GEMM dispatches in pass A")
(cB, nB), (cC, nC) = counters("B"), counters("C")
for k in ("attn_kernel", "gemm_decode64_kernel", "gemm_decode_ln_kernel", "ln_kernel"):
    if nB[k] == 0 and dT[k][1] == 0: continue
    f_kb, w_kb, ns = cB[k]["FETCH_SIZE"], cC[k]["WRITE_SIZE"], dT[k][0]
    if nB[k] == 0 or ns == 0 or f_kb == 0: problems.append(f"{k}: dispatches={nB[k]} duration={ns} fetch={f_kb}")
    res[k] = {"dispatches": dT[k][1], "duration_ns": ns, "avg_us": ns / 1e3 / max(1, dT[k][1]), "FETCH_SIZE_kb_raw": f_kb, "WRITE_SIZE_kb": w_kb,
              "hbm_bytes": (2 * f_kb + w_kb) * 1024.0, "hbm_GBps": (2 * f_kb + w_kb) * 1024.0 / ns if ns else None,
              "hbm_GBps_uncorrected": (f_kb + w_kb) * 1024.0 / ns if ns else None}
for k in ("attn_kernel", "gemm_decode64_kernel"):
    if k not in res: problems.append(f"{k}: not dispatched")
if problems:
    print("pmc_gpt: INVALID RUN:", "; ".join(problems), file=sys.stderr)
    sys.exit(1)
# decode step as a whole: bytes of every decode kernel / their summed durations
dec = [k for k in ("attn_kernel", "gemm_decode64_kernel", "gemm_decode_ln_kernel", "ln_kernel") if k in res]
tot_b, tot_ns = sum(res[k]["hbm_bytes"] for k in dec), sum(res[k]["duration_ns"] for k in dec)
res["decode_kernels_total"] = {"kernels": dec, "hbm_bytes": tot_b, "duration_ns": tot_ns, "hbm_GBps": tot_b / tot_ns,
                               "note": "prefill attention / LayerNorm launches of the one prefill pass are in these sums too (< 5 % at 24 tokens)"}
json.dump(res, open(f"{out}/gpt_pmc_b{B}.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rc=$?
rm -f "$OUT"/cc_*.csv "$OUT"/kt_*.csv
exit $rc
