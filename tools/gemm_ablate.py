#!/usr/bin/env python3
"""Decode-GEMM microbenchmark with ablations (GPU box): where do the ~12 us per weight-streaming GEMM go?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indextts_amd import gpt, _lib

dev = "cuda:0"
L = _lib.lib()
def run(M, K, N, ablate, iters=300):
    a = torch.randn(M, K).bfloat16().to(dev)
    w = torch.randn(K, N) / K ** 0.5
    wp = gpt.pack_gemm_weight(w, 1).to(dev)
    # many distinct weight copies so the stream comes from HBM, not L2/MALL
    copies = [wp.clone() for _ in range(max(1, int(600e6 // wp.numel())))]
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev)
    st = _lib.stream_ptr()
    def call(i):
        L.itts_gemm_forward(_lib.ptr(a), _lib.ptr(copies[i % len(copies)]), _lib.ptr(bias), _lib.ptr(out), M, N, K, 1, 0, ablate, st)
    for i in range(10): call(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): call(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for (K, N) in [(1280, 3840), (1280, 5120), (5120, 1280), (1280, 1280), (1280, 8194)]:
    for M in (8, 64):
        res = {ab: run(M, K, N, ab) for ab in (0, 1, 2, 3, 4, 7)}
        mb = K * N * 2 / 1e6
        print(f"K={K} N={N} M={M} ({mb:.1f} MB): " + "  ".join(f"ab{ab}={t:6.2f}us" for ab, t in res.items()) + f"   -> {mb / res[0] * 1e3 / 1e3:.2f} TB/s", flush=True)
