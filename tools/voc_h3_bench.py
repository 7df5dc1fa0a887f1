#!/usr/bin/env python3
"""GPU box: BigVGAN forward, exact-f32 conv mode vs the opt-in f16 x 3 split-operand mode: wall ms per forward and per-stage conv time
(HIP-event records; in f16x3 mode a conv record covers the split pass + the conv kernel)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indextts_amd import _lib, bigvgan, synth  # noqa: E402

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 1926
MODES = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f32", "f16x3:96", "f16x3:192"]
bh = dict(synth.BIGVGAN_V2_22K)
sd = synth.bigvgan_weights(bh)
mel = (torch.randn(B, 80, T, generator=torch.Generator().manual_seed(0)) * 2 - 4).cuda()
ref = None
for spec in MODES:
    mode, _, rest = spec.partition(":")                      # mode[:min_channels[:option=value[+option=value...]]]
    minc, _, opts = rest.partition(":")
    _lib.reset_options()
    for kv in filter(None, opts.split("+")):
        _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    voc = bigvgan.BigVGAN(bh, conv_mode=mode, h3_min_channels=int(minc or 0))
    voc.load_state_dict(sd)
    voc.to("cuda:0")
    wav = voc(mel)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2):
        wav = voc(mel)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 2
    voc.set_profiling(True)
    voc(mel)
    recs = voc.profile_records()
    i, per_stage = 1, []
    for st in range(6):
        i += 1
        conv_ms = conv_fl = act_ms = 0.0
        for _ in range(9):
            for nm in range(4):
                cls, t, fl, by = recs[i]
                i += 1
                if cls == 0:
                    conv_ms += t
                    conv_fl += fl
                else:
                    act_ms += t
        per_stage.append(f"s{st}: conv {conv_ms:7.2f} ms {conv_fl / conv_ms / 1e9:6.1f} TF | act {act_ms:6.2f}")
    w = wav.float().cpu()
    if ref is None:
        ref = w
    d = float((w - ref).pow(2).mean().sqrt())
    print(f"B={B} T={T} mode={spec}: {ms:8.2f} ms / forward   rms diff vs first mode {d:.2e} (signal {float(ref.pow(2).mean().sqrt()):.3f})")
    for s in per_stage:
        print("   ", s)
    del voc
    torch.cuda.empty_cache()
