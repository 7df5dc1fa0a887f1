#!/bin/bash
# GPU box: HBM traffic of the dominant kernel from PMC counters (separate passes for FETCH_SIZE and WRITE_SIZE, as
# MI355X_MICROARCH.md prescribes), calibrated on a kernel with a known byte count in the same access width
# (aa_act_kernel: reads N*4 B and writes N*4 B with coalesced dword accesses on a tensor larger than the 256 MiB MALL).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc
mkdir -p "$OUT"
cat > /tmp/pmc_driver.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from indextts_amd import bigvgan as bv
dev = "cuda:0"
# calibration: activation on 2 x 768 x 262144 floats = 1.61 GB in, 1.61 GB out
x = torch.randn(2, 768, 262144, device=dev)
al = torch.zeros(768, device=dev); f = torch.ones(12, device=dev) / 12
for _ in range(2): y = bv.anti_alias_activation(x, f, f, al, al)
torch.cuda.synchronize(); del x, y
# the dominant kernel at the bench shape of stage 1: C=768, k=7, d=3, B=64... use B=16 (tensor 378 MB > MALL)
C, k, d, T, B = 768, 7, 3, 7704, 16
xx = torch.randn(B, C, T, device=dev)
w = torch.randn(C, C, k) / (C * k) ** 0.5
wp = bv.pack_conv1d_weight(w).to(dev); bias = torch.zeros(C, device=dev); out = torch.empty_like(xx)
for _ in range(2): bv.conv1d(xx, wp, bias, C, k, d, out=out)
torch.cuda.synchronize()
print("algorithmic: act bytes in/out", 2*768*262144*4, "conv x bytes", xx.numel()*4, "y bytes", out.numel()*4, "w bytes", w.numel()*4, "flops", 2.0*C*C*k*T*B)
PY
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/raw_$ctr" -o p -- python /tmp/pmc_driver.py > "$OUT/run_$ctr.log" 2>&1
  f=$(find "$OUT/raw_$ctr" -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$ctr" <<'PY' | tee -a "$OUT/summary.txt"
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") == ctr:
        agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "aa_act" in k or "conv_mfma" in k:
        print(f"{ctr} {k}: per-dispatch values {[round(x, 1) for x in v]}")
PY
  rm -rf "$OUT/raw_$ctr"
done
grep algorithmic "$OUT/run_FETCH_SIZE.log" | tee -a "$OUT/summary.txt"
