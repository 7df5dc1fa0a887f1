"""GPU box: prefill GEMM TFLOP/s at the bench shapes (M = 64 x 135 rows), both kernels (ITTS_PREFILL_GEMM=0/1 per process)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import gpt  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8640
dev = "cuda:0"
for (K, N, name) in ((1280, 3840, "qkv"), (1280, 1280, "proj"), (1280, 5120, "fc1"), (5120, 1280, "fc2")):
    g = torch.Generator().manual_seed(K + N)
    a = (torch.randn(M, K, generator=g)).bfloat16().to(dev)
    w = torch.randn(K, N, generator=g) / K ** 0.5
    wp = gpt.pack_gemm_weight(w, 1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    y = gpt.gemm(a, wp, bias, N, 1, prefill_tiles=True)
    ref = a.float() @ w.bfloat16().float().to(dev) + bias
    err = float((y - ref).abs().max() / ref.abs().max())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gpt.gemm(a, wp, bias, N, 1, prefill_tiles=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"ITTS_PREFILL_GEMM={os.environ.get('ITTS_PREFILL_GEMM', '1')} {name}: M={M} K={K} N={N}  {ms * 1e3:.1f} us  "
          f"{2.0 * M * K * N / ms / 1e9:.1f} TFLOP/s  rel.err {err:.2e}", flush=True)
