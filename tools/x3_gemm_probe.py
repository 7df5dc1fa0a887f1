"""GPU box: the x3 GEMM kernel variants (plane products x split/MFMA interleave) at a solve-sized M: run-to-run determinism and error vs f64,
with the location pattern of any element that is off.  usage: x3_gemm_probe.py [M] [N] [K]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import _lib, gpt  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (39088, 1536, 512)
g = torch.Generator().manual_seed(5)
a = torch.randn(M, K, generator=g)
w = torch.randn(K, N, generator=g) / K ** 0.5
b = torch.randn(N, generator=g)
ref = a.double() @ w.double() + b.double()
ad, bd = a.cuda(), b.cuda()
hsh = lambda y: hashlib.sha1(y.numpy().tobytes()).hexdigest()[:10]
for prec, prods, sched in ((0, None, None), (2, 8, 1), (2, 8, 0), (2, 6, 1), (2, 6, 0)):
    opts = {} if prods is None else {"x3_products": prods, "x3_sched": sched}
    with _lib.option_scope(**opts):
        wp = gpt.pack_gemm_weight(w, prec).cuda()
        ys = [gpt.gemm(ad, wp, bd, N, prec, prefill_tiles=True).cpu() for _ in range(4)]
    d = (ys[0].double() - ref).abs()
    bad = (d > 1e-4).nonzero()
    print(f"GEMM {M}x{N}x{K} prec {prec} products {prods} sched {sched}: hashes {[hsh(y) for y in ys]} max|d| vs f64 {float(d.max()):.3e} "
          f"rms {float(d.pow(2).mean().sqrt()):.3e}; elements off by > 1e-4: {len(bad)}", flush=True)
    if len(bad):
        rows = bad[:, 0]
        print("   rows % 128:", sorted(set((rows % 128).tolist()))[:40], " cols % 128:", sorted(set((bad[:, 1] % 128).tolist()))[:40],
              " row tiles:", sorted(set((rows // 128).tolist()))[:20], " first:", bad[:6].tolist(), flush=True)
