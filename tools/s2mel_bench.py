"""GPU box: time the s2mel CFG Euler solve at the shipped widths (hidden 512 x 13 layers x 8 heads, WaveNet 512 x 8) on a packed
batch.  usage: s2mel_bench.py [n_utts] [prompt_frames] [gen_frames] [steps] [precision[:opt=v,opt=v] ...]
(several precision specs are timed one after the other on the same inputs; opt = an engine option, e.g. fp32x3:x3_attn=0,x3_products=8)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import s2mel, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Tp = int(sys.argv[2]) if len(sys.argv) > 2 else 800
Tg = int(sys.argv[3]) if len(sys.argv) > 3 else 1926
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 25
specs = sys.argv[5:] if len(sys.argv) > 5 else ["bf16"]
args = synth.S2MEL_V2
g = torch.Generator().manual_seed(0)
T = Tp + Tg
x = torch.randn(B, 80, T, generator=g).cuda()
mu = torch.randn(B, T, args["DiT"]["content_dim"], generator=g).cuda()
prompt = (torch.randn(1, 80, Tp, generator=g) * 0.5 - 1.0).cuda()
style = torch.randn(1, args["style_encoder"]["dim"], generator=g).cuda()
t_span = torch.linspace(0, 1, steps + 1)
lens = torch.full((B,), T)
from indextts_amd import _lib  # noqa: E402
H, I, W, L, D = 512, 1536, 512, 8, 13
tok = 2 * B * T
gemm = tok * (D * 2 * (H * 3 * H + H * H + H * 2 * I + I * H) + (D // 2) * 2 * 2 * H * H + L * 2 * (5 * W * 2 * W + W * 2 * W) + 2 * (3 * H * W + W * W))
attn = 2 * B * D * 4 * T * T * H
y_first = None
for spec in specs:
    prec, _, optstr = spec.partition(":")
    opts = {k: int(v) for k, v in (kv.split("=") for kv in optstr.split(",") if kv)}
    with _lib.option_scope(**opts):
        m = s2mel.CFM(args, precision=prec, device="cuda:0")
        m.load_state_dict(synth.s2mel_weights(args, seed=1234))
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y = m.solve_euler(x.clone(), lens, prompt, mu, style, None, t_span, 0.7, frame_lens=[T] * B)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    del m
    if y_first is None:
        y_first = y
    print(f"B={B} T={T} steps={steps} {spec}: {dt * 1e3:.1f} ms total, {dt / steps * 1e3:.2f} ms/step; "
          f"GEMM {gemm * steps / 1e12:.1f} TFLOP + attention {attn * steps / 1e12:.1f} TFLOP -> {(gemm + attn) * steps / dt / 1e12:.0f} TFLOP/s; "
          f"finite={bool(torch.isfinite(y).all())} rms={float(y.pow(2).mean().sqrt()):.3f} max|d| vs the first spec {float((y - y_first).abs().max()):.3e} (rms {float((y - y_first).pow(2).mean().sqrt()):.3e}) "
          f"bits={__import__('hashlib').sha1(y.float().cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
