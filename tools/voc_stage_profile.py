#!/usr/bin/env python3
"""Per-launch BigVGAN timing (HIP events) grouped by stage/kernel size: where does the vocoder time go."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indextts_amd import bigvgan, synth

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 1926
bh = dict(synth.BIGVGAN_V2_22K)
voc = bigvgan.BigVGAN(bh); voc.load_state_dict(synth.bigvgan_weights(bh)); voc.to("cuda:0"); voc.set_profiling(True)
mel = (torch.randn(B, 80, T) * 2 - 4).cuda()
voc(mel); voc(mel)
recs = voc.profile_records()
names = {0: "conv", 1: "convT", 2: "act", 3: "post"}
# launch order: conv_pre, then per stage: convT, then per resblock j (k=3,7,11), per dilation: act, conv1, act, conv2
rows = []
i = 0
rows.append(("conv_pre", recs[0])); i = 1
for st in range(6):
    rows.append((f"s{st} up", recs[i])); i += 1
    for j, k in enumerate((3, 7, 11)):
        for d in range(3):
            for nm in ("act1", "conv1", "act2", "conv2"):
                rows.append((f"s{st} k{k} d{d} {nm}", recs[i])); i += 1
agg = {}
for name, (cls, ms, fl, by) in rows:
    parts = name.split()
    key = (parts[0], parts[1] if len(parts) > 1 and parts[1].startswith("k") else "", names[cls])
    a = agg.setdefault(key, [0.0, 0.0, 0.0, 0]); a[0] += ms; a[1] += fl; a[2] += by; a[3] += 1
tot = sum(r[1][1] for r in rows)
print(f"B={B} T={T} total {tot:.1f} ms")
for key, (ms, fl, by, n) in agg.items():
    print(f"{key[0]:9s} {key[1]:4s} {key[2]:6s} n={n:2d} {ms:8.2f} ms {100*ms/tot:5.1f}%  {fl/ms/1e9 if ms else 0:7.1f} TF  {by/ms/1e6 if ms else 0:8.1f} GB/s")
