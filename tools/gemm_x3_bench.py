"""GPU box: time the s2mel GEMM shapes through itts_gemm_forward in the native f32-MFMA mode (precision 0) and the f32x3 mode (2).
usage: gemm_x3_bench.py [M] [reps]   (plain store epilogue; M = packed rows, default 39088 = 8 x 2 x 2443)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import gpt  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 39088
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = torch.Generator().manual_seed(0)
for N, K in ((1536, 512), (512, 512), (3072, 512), (512, 1536), (1024, 2560)):
    a = torch.randn(M, K, generator=g).cuda()
    w = torch.randn(K, N, generator=g) / K ** 0.5
    for prec, name in ((0, "f32"), (2, "f32x3")):
        wp = gpt.pack_gemm_weight(w, prec).cuda()
        gpt.gemm(a, wp, None, N, prec, prefill_tiles=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            gpt.gemm(a, wp, None, N, prec, prefill_tiles=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"M={M} N={N} K={K} {name}: {dt * 1e3:.3f} ms  {2.0 * M * N * K / dt / 1e12:.1f} TFLOP/s (f32-equivalent)", flush=True)
