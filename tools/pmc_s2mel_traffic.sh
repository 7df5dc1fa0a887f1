#!/bin/bash
# usage: tools/pmc_s2mel_traffic.sh [B=64] [precision=fp32x3|fp32]
# GPU box: HBM traffic of the s2mel GEMM launches (gemm_x3_kernel in the fp32x3 mode / gemm_prefill_kernel<..., F32 = true> in fp32: the dominant kernel of the step) over
# ONE Euler step (one CFG-stacked estimator call) at the bench shape, from PMC counters in two separate rocprofv3 passes (FETCH_SIZE,
# WRITE_SIZE; MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced 16 B / lane reads --
# LDS-DMA included -- so it is doubled; WRITE_SIZE measured 1.000 on a known byte count in round 2, profiles/conv_traffic.json).
set -u
B=${1:-64}
PREC=${2:-fp32x3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_s2mel
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/raw_$ctr" -o p -- python $ROOT/tools/s2mel_bench.py $B 517 1926 1 $PREC > "$OUT/run_$ctr.log" 2>&1
  grep -q "finite=True" "$OUT/run_$ctr.log" || { echo "pmc_s2mel_traffic: workload of pass $ctr failed: $(tail -3 "$OUT/run_$ctr.log")" >&2; exit 1; }
  f=$(find "$OUT/raw_$ctr" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|gemm_prefill_kernel|gemm_x3|flash_attn_f32|flash_attn_x3" "$f" > "$OUT/cc_$ctr.csv"
  rm -rf "$OUT/raw_$ctr"
done
python3 - "$OUT" "$B" "$PREC" <<'PY'
import csv, json, sys, collections
out, B, PREC = sys.argv[1], int(sys.argv[2]), sys.argv[3]
T, Tp = 1926, 517
M = 2 * B * (T + Tp)                       # packed rows: CFG branches x utterances x frames
H, I, W, C, Kx, L, D = 512, 1536, 512, 80, 128, 8, 13
f4 = 4.0
def gemm(N, K, out_elems_per_row, extra_read_per_row=0):      # algorithmic bytes of one launch: A + W once + outputs (+ read-modify-write input)
    return M * K * f4 + K * N * f4 + M * (out_elems_per_row + extra_read_per_row) * f4
alg = []
alg.append(gemm(H, Kx, H, H))                                 # x columns of cond_x_merge_linear, residual onto the constant part
for i in range(D):
    if i > D // 2:
        alg += [gemm(H, H, H), gemm(H, H, H, H)]              # skip_in_linear halves (store, then residual)
    alg.append(gemm(3 * H, H, 3 * H))                         # wqkv -> Q, K, V^T
    alg.append(gemm(H, H, H, H))                              # wo residual
    alg.append(gemm(2 * I, H, I))                             # w1 | w3 -> SwiGLU
    alg.append(gemm(H, I, 2 * H, H))                          # w2 residual + f32 shadow
alg += [gemm(H, H, H), gemm(H, Kx, 2 * H, H), gemm(W, H, 2 * W), gemm(W, H, W)]      # skip_linear halves, conv1 (+ shadow), res_projection
for i in range(L):
    alg.append(gemm(2 * W, 5 * W, W))                         # tap-mode dilated conv + gate (A rows re-read across taps come from L2)
    last = i == L - 1
    alg.append(gemm(W if last else 2 * W, W, (1 if last else 3) * W, (1 if last else 2) * W))     # res / skip: RMW of x and skip sum (+ shadow)
alg += [gemm(W, W, 2 * W), gemm(C, W, C)]                     # final_layer.linear (+ shadow), conv2
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f"{out}/cc_{ctr}.csv")):
        if r.get("Counter_Name") == ctr:
            k = "gemm" if ("gemm_prefill_kernel" in r["Kernel_Name"] or "gemm_x3" in r["Kernel_Name"]) else "attn"
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    res[ctr] = {k: {"kb_sum": v[0], "dispatches": v[1]} for k, v in agg.items()}
n = res["FETCH_SIZE"].get("gemm", {}).get("dispatches", 0)
if n == 0:
    print("pmc_s2mel_traffic: no GEMM dispatches in the counter file", file=sys.stderr); sys.exit(1)
fetch = res["FETCH_SIZE"]["gemm"]["kb_sum"] * 1024.0 * 2.0      # gfx950: half of the coalesced 16 B / lane bytes are reported
write = res["WRITE_SIZE"]["gemm"]["kb_sum"] * 1024.0
summary = {"B": B, "mel_frames": T, "prompt_frames": Tp, "precision": PREC, "rows": M, "gemm_dispatches": n, "expected_gemm_launches": len(alg),
           "fetch_kb_sum_raw": res["FETCH_SIZE"]["gemm"]["kb_sum"], "write_kb_sum_raw": res["WRITE_SIZE"]["gemm"]["kb_sum"],
           "fetch_correction": 2.0, "write_correction": 1.0,
           "hbm_bytes_per_gemm_launch": (fetch + write) / max(1, n), "algorithmic_bytes_per_gemm_launch": sum(alg) / len(alg),
           "attention": {k: res[c].get("attn") for k, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"))},
           "source": "tools/pmc_s2mel_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over one Euler step at the bench shape; "
                     "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B / lane coalesced reads on gfx950"}
json.dump(summary, open(f"{out}/s2mel_gemm_traffic.json", "w"), indent=1)
print(json.dumps(summary))
PY
