#!/usr/bin/env python3
"""Quick per-kernel timing on the GPU box (not the judged bench): conv / activation throughput."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indextts_amd import bigvgan as bv  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    out = []
    for (C, k, d, T, B) in [(768, 3, 1, 7704, 8), (768, 7, 3, 7704, 8), (768, 11, 5, 7704, 8), (384, 7, 1, 30816, 4),
                            (192, 11, 1, 61632, 2), (96, 7, 1, 123264, 2), (48, 7, 1, 246528, 1), (24, 7, 1, 493056, 1)]:
        x = torch.randn(B, C, T, device=DEV)
        w = torch.randn(C, C, k) / (C * k) ** 0.5
        wp = bv.pack_conv1d_weight(w).to(DEV)
        bias = torch.zeros(C, device=DEV)
        y = torch.empty_like(x)
        ms = timeit(lambda: bv.conv1d(x, wp, bias, C, k, d, out=y))
        fl = 2.0 * C * C * k * T * B
        al = torch.zeros(C, device=DEV)
        f = torch.ones(12, device=DEV) / 12
        ms_a = timeit(lambda: bv.anti_alias_activation(x, f, f, al, al))
        rec = dict(C=C, k=k, d=d, T=T, B=B, conv_ms=ms, conv_tflops=fl / ms / 1e9, act_ms=ms_a,
                   act_GBps=2 * x.numel() * 4 / ms_a / 1e6)
        print(json.dumps(rec))
        out.append(rec)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_kernels.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
