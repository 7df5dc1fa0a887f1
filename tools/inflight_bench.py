#!/usr/bin/env python3
"""GPU box: the GPT stage of N ragged utterances on S decode slots -- in flight (freed slots refilled from the waiting utterances,
UnifiedVoice.inference_speech_inflight) against drained batches of S (row compaction on in both).  Production GPT widths, synthetic weights, EOS
suppressed: every utterance runs to its cap.  usage: inflight_bench.py [N=128] [slots=64] [chunk_tokens=32] [min_free=8] [lo=120] [hi=560]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indextts_amd import gpt, synth  # noqa: E402

N, S, CHUNK, MINFREE, LO, HI = [int(sys.argv[i + 1]) if len(sys.argv) > i + 1 else d for i, d in enumerate((128, 64, 32, 8, 120, 560))]
dev = "cuda:0"
cfg = dict(synth.GPT_V25)
m = gpt.UnifiedVoice(**cfg, spk_cond_mode="campplus", precision="bf16", device=dev)
m.load_state_dict(synth.gpt_weights(cfg, seed=1234, suppress_eos=True))
m.post_init_gpt2_config(kv_cache=True, half=True)
m.set_compaction(True, 8)
g = torch.Generator().manual_seed(308)
caps = torch.randint(LO, HI + 1, (N,), generator=g).tolist()
text = torch.cat([torch.randint(2, 12000, (N, 128), generator=g).to(torch.int32), torch.ones(N, 1, dtype=torch.int32)], dim=1).to(dev)
langs = torch.full((N,), 3, dtype=torch.long, device=dev)
style = (torch.randn(1, 192, generator=g) * 0.1).to(dev)
emo = (torch.randn(1, cfg["model_dim"], generator=g) * 0.1).to(dev)
kw = dict(emo_vec=emo, campplus_embedding=style, max_generate_length=HI, do_sample=True, num_beams=1, top_p=0.8, top_k=30, temperature=0.8,
          repetition_penalty=10.0)


def drained():
    return [m.inference_speech(None, text[i:i + S], langs=langs[i:i + S], row_max_new=caps[i:i + S], **kw)[0] for i in range(0, N, S)]


def inflight():
    return m.inference_speech_inflight(None, text, langs=langs, slots=S, chunk_tokens=CHUNK, min_free=MINFREE, row_max_new=caps, **kw)[0]


def wall(f):
    best, res = None, None
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = f()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, res


def lens_of(c):
    return [int((r == m.stop_mel_token).nonzero()[0]) if bool((r == m.stop_mel_token).any()) else int(r.numel()) for r in c]


t_d, c_d = wall(drained)
t_i, c_i = wall(inflight)
ok = [n for c in c_d for n in lens_of(c)] == caps and lens_of(c_i) == caps
tok = sum(caps)
print(f"N={N} slots={S} chunk={CHUNK} min_free={MINFREE} caps {LO}..{HI} (mean {tok / N:.0f}): drained {t_d:.3f} s ({tok / t_d:.0f} tokens/s)  "
      f"in flight {t_i:.3f} s ({tok / t_i:.0f} tokens/s)  speedup {t_d / t_i:.3f}  lengths as capped: {ok}  schedule {m.last_inflight}")
