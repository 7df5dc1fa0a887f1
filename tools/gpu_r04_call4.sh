#!/bin/bash
# round 4, GPU call 4: why does the 8-product x3 solve sit 2-4e-3 from the f32 solve at 8 x 2443 frames when the 6-product one sits at 8e-6?
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r04d
mkdir -p $O
timeout 300 python tools/s2mel_bench.py 8 517 1926 1 fp32 fp32x3:x3_attn=0,x3_products=8 fp32x3:x3_attn=0,x3_products=8,x3_sched=0 fp32x3:x3_attn=0,x3_products=8 fp32x3:x3_attn=0,x3_products=6,x3_sched=0 > $O/dbg_b8.log 2>&1; echo "b8 rc=$?"
grep "^B=" $O/dbg_b8.log
timeout 300 python tools/s2mel_bench.py 2 517 1926 1 fp32 fp32x3:x3_attn=0,x3_products=8 fp32x3:x3_attn=0,x3_products=8,x3_sched=0 fp32x3:x3_attn=0,x3_products=6 > $O/dbg_b2.log 2>&1; echo "b2 rc=$?"
grep "^B=" $O/dbg_b2.log
timeout 300 python - > $O/dbg_gemm.log 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, ".")
from indextts_amd import _lib, gpt
g = torch.Generator().manual_seed(5)
for (M, N, K) in [(39088, 1536, 512), (39088, 512, 1536), (39088, 512, 512), (9720, 1536, 512)]:
    a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    ref = (a.double() @ w.double().t() + b.double())
    for prec, prods in (("fp32", None), ("fp32x3", 8), ("fp32x3", 6)):
        with _lib.option_scope(**({"x3_products": prods} if prods else {})):
            wp = gpt.pack_gemm_weight(w, prec).to("cuda:0")
            y = gpt.gemm(a.to("cuda:0"), wp, b.to("cuda:0"), N, prec, prefill_tiles=True).cpu().double()
        d = y - ref
        bad = (d.abs() > 1e-4).nonzero()
        print(f"GEMM {M}x{N}x{K} {prec} {prods}: max|d| {float(d.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e} bad {len(bad)} first {bad[:3].tolist()}", flush=True)
PY
echo "gemm rc=$?"; cat $O/dbg_gemm.log | tail -14
