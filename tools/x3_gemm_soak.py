"""GPU box: sustained run of ONE GEMM variant on fixed inputs; every output is compared on the device with the first one.
usage: x3_gemm_soak.py M N K reps prec[:opt=v,...] ..."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import _lib, gpt  # noqa: E402

M, N, K, reps = (int(v) for v in sys.argv[1:5])
g = torch.Generator().manual_seed(5)
a = torch.randn(M, K, generator=g).cuda()
w = torch.randn(K, N, generator=g) / K ** 0.5
b = torch.randn(N, generator=g).cuda()
PREC = {"fp32": 0, "bf16": 1, "fp32x3": 2}
for spec in sys.argv[5:]:
    name, _, optstr = spec.partition(":")
    opts = {k: int(v) for k, v in (kv.split("=") for kv in optstr.split(",") if kv)}
    prec = PREC[name]
    with _lib.option_scope(**opts):
        wp = gpt.pack_gemm_weight(w, prec).cuda()
        aa = a.bfloat16() if prec == 1 else a
        y0 = gpt.gemm(aa, wp, b, N, prec, prefill_tiles=True)
        nbad, worst = 0, 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bad_iters = []
        for i in range(reps):
            y = gpt.gemm(aa, wp, b, N, prec, prefill_tiles=True)
            if i % 8 == 7 or i == reps - 1:
                ne = int((y != y0).sum())
                if ne:
                    nbad += 1
                    worst = max(worst, float((y - y0).abs().max()))
                    if len(bad_iters) < 5:
                        bad_iters.append((i, ne, (y != y0).nonzero()[:3].tolist()))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"GEMM {M}x{N}x{K} {spec}: {reps} launches in {dt:.2f} s ({2.0 * M * N * K * reps / dt / 1e12:.0f} TFLOP/s incl. checks); "
          f"checked outputs that differ from the first: {nbad} (worst |d| {worst:.3e}) {bad_iters}", flush=True)
