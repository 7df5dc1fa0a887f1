"""GPU box: time the s2mel GEMM shapes through itts_gemm_forward in the native f32-MFMA mode (precision 0) and the f32x3 mode (2).
usage: gemm_x3_bench.py [M] [reps] [option=value ...]   (plain store epilogue; M = packed rows, default 39088 = 8 x 2 x 2443; the options are engine
options set for the whole run, e.g. x3_prio=1)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import gpt, _lib  # noqa: E402

_pos = [v for v in sys.argv[1:] if "=" not in v]
_opts = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[1:] if "=" in kv)}
M = int(_pos[0]) if len(_pos) > 0 else 39088
reps = int(_pos[1]) if len(_pos) > 1 else 5
for _k, _v in _opts.items():
    _lib.set_option(_k, _v)
if _opts:
    print("options:", _opts, flush=True)
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
