"""Helper for test_gpu_gpt.py::test_fused_layernorm_decode_steps_equal_unfused: decode loops of 1-4 rows on the bf16 engine, digest of the ids.
PROBE_OPTS ("name=value,...": engine options applied through itts_set_option before the model is built) selects the kernels under test,
one process per setting."""
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

BIG = os.environ.get("PROBE_BIG") == "1"
cfg = (G.GPTConfig(layers=3, model_dim=1280, heads=20, max_text_tokens=60, max_mel_tokens=80, number_text_tokens=200) if BIG
       else G.GPTConfig(layers=3, model_dim=256, heads=4, max_text_tokens=60, max_mel_tokens=80, number_text_tokens=200))
sd = G.synth_weights(cfg, seed=78)
m = gpt.UnifiedVoice(spk_cond_mode="campplus", layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, max_text_tokens=cfg.max_text_tokens,
                     max_mel_tokens=cfg.max_mel_tokens, number_text_tokens=cfg.number_text_tokens, precision="bf16", device="cuda:0")
m.load_state_dict(sd)
g = torch.Generator().manual_seed(6)
style = torch.randn(1, 192, generator=g)
emo = torch.randn(1, cfg.model_dim, generator=g) * 0.1
h = hashlib.sha256()
first = None
for B in (1, 3, 4, 5, 8, 11, 16, 17):                    # 1-4: one row per wave on 4 waves; 5-8: on 8 waves; 9-16: two rows per wave; 17: the unfused path
    text = torch.randint(2, 200, (B, 23), generator=g)
    if B > 1:
        text[1, 11:] = 1
    langs = torch.full((B,), 2)
    for kw in (dict(do_sample=False, num_beams=1), dict(do_sample=True, num_beams=1, top_p=0.8, top_k=30, temperature=0.8, seed=11),
               dict(do_sample=True, num_beams=3, top_p=0.8, top_k=30, temperature=0.8, seed=12, length_penalty=0.0)):
        if kw["num_beams"] == 3 and B not in (1, 4):             # 3 and 12 rows
            continue
        codes, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=40,
                                      repetition_penalty=10.0, **kw)
        h.update(codes.cpu().numpy().tobytes())
        first = first if first is not None else codes[0, :8].tolist()
print("DIGEST", h.hexdigest(), first)
