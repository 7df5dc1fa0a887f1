"""Helper for test_gpu_gpt.py::test_prefill_gemm_kernels_agree: runs the bf16 engine's prefill-shaped paths (teacher-forced
latent pass + a short greedy decode) and prints a digest of the raw outputs.  PROBE_OPTS ("name=value,...": engine options applied through
itts_set_option) selects the GEMM kernels under test, one process per setting."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import gpt  # noqa: E402
from oracle import gpt_oracle as G  # noqa: E402  (seeded synthetic weights only)
from indextts_amd import _lib  # noqa: E402
for _kv in filter(None, os.environ.get("PROBE_OPTS", "").split(",")):      # engine options of this run (itts_set_option), e.g. "decode_fuse_ln=0"
    _lib.set_option(_kv.split("=")[0], int(_kv.split("=")[1]))

BIG = os.environ.get("PROBE_BIG") == "1"       # full-width stack (K = 1280 / 5120 slices) and 40 rows: the 33..64-row decode GEMM
cfg = (G.GPTConfig(layers=2, model_dim=1280, heads=20, max_text_tokens=60, max_mel_tokens=80, number_text_tokens=200) if BIG
       else G.GPTConfig(layers=3, model_dim=256, heads=4, max_text_tokens=60, max_mel_tokens=80, number_text_tokens=200))
DIM = int(os.environ.get("PROBE_DIM", "0"))    # odd multiples of 128: split-K slices with an odd number of 32-wide k-blocks
if DIM:
    cfg = G.GPTConfig(layers=2, model_dim=DIM, heads=DIM // 64, max_text_tokens=60, max_mel_tokens=80, number_text_tokens=200)
sd = G.synth_weights(cfg, seed=77)
m = gpt.UnifiedVoice(spk_cond_mode="campplus", layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, max_text_tokens=cfg.max_text_tokens,
                     max_mel_tokens=cfg.max_mel_tokens, number_text_tokens=cfg.number_text_tokens, precision="bf16",
                     device="cuda:0")
m.load_state_dict(sd)
g = torch.Generator().manual_seed(5)
B = int(os.environ.get("PROBE_B", "40" if BIG else "5"))
text = torch.randint(2, 200, (B, 37), generator=g)
text[1, 20:] = 1
text[3, 9:] = 1
tl = torch.full((B,), 37)
tl[1], tl[3] = 20, 9
mel = torch.randint(0, 8192, (B, 61), generator=g)
ml = torch.full((B,), 61)
ml[1], ml[2], ml[4] = 40, 13, 2
style = torch.randn(1, 192, generator=g)
emo = torch.randn(1, cfg.model_dim, generator=g) * 0.1
conds, _ = m.conds_latent(style, emo)
lat = m.forward_latent(conds.repeat(B, 1, 1), text, tl, mel, ml)           # M = 5 * (3 + 39 + 63) = 525 rows: 5 m-tiles, ragged last
codes, _ = m.inference_speech(None, text, langs=torch.full((B,), 2), emo_vec=emo, campplus_embedding=style,
                              max_generate_length=12, do_sample=False, num_beams=1, repetition_penalty=10.0)
h = hashlib.sha256()
h.update(lat.float().cpu().numpy().tobytes())
h.update(codes.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest(), float(lat.float().abs().mean()), tuple(lat.shape), codes[0, :6].tolist())
