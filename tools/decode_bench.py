"""GPU box: GPT decode cost per token (bf16, sampled, num_beams = 1, hipGraph) at several batch sizes and engine-option settings, one process.
usage: decode_bench.py <n_gen> <B,B,...> [name=value[,name=value] ...]   -- each option group is timed at every B next to the defaults
e.g.   decode_bench.py 560 1,8,16,64 attn_waves=4 attn_waves=16 decode_fuse_ln=0"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import _lib, gpt, synth  # noqa: E402

n_gen = int(sys.argv[1]) if len(sys.argv) > 1 else 560
Bs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,8,64").split(",")]
groups = [{}] + [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) for a in sys.argv[3:]]
gcfg = dict(synth.GPT_V25)
m = gpt.UnifiedVoice(spk_cond_mode="campplus", **gcfg, precision="bf16", device="cuda:0")
m.load_state_dict(synth.gpt_weights(gcfg, seed=1234, suppress_eos=True))
m.post_init_gpt2_config(kv_cache=True, half=True)
g = torch.Generator().manual_seed(0)
style = torch.randn(1, 192, generator=g).cuda()
emo = (torch.randn(1, 1280, generator=g) * 0.1).cuda()
kw = dict(do_sample=True, top_p=0.8, top_k=30, temperature=0.8, num_beams=1, repetition_penalty=10.0, length_penalty=0.0)
D, L, V = gcfg["model_dim"], gcfg["layers"], gcfg["number_mel_codes"]
for B in Bs:
    text = torch.randint(2, 12000, (B, 128), generator=torch.Generator().manual_seed(B)).cuda()
    langs = torch.full((B,), 3, dtype=torch.long).cuda()
    ref = None
    for opts in groups:
        with _lib.option_scope(**opts):
            best = None
            for rep in range(3):
                codes, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=n_gen, seed=7, **kw)
                t = m.last_timing
                ms = t["decode_ms"] / max(1, t["steps"] - 1)
                best = ms if best is None or (rep > 0 and ms < best) else best
        same = "" if ref is None else (" ids==default" if torch.equal(codes, ref) else " IDS DIFFER FROM DEFAULT")
        ref = codes if ref is None else ref
        ctx = 134 + n_gen / 2.0
        gb = ((12 * D * D * L + D * V) * 2 + B * 2 * L * D * ctx * 2) / 1e9
        print(f"B={B:3d} n={n_gen} opts={opts or 'default'}: {best:.4f} ms/token  ({gb / best:.2f} TB/s algorithmic = {gb / best / 8 * 100:.1f} % of 8 TB/s)"
              f" prefill {t['prefill_ms']:.1f} ms{same}", flush=True)
