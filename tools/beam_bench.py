"""GPU box: decode cost per token of the beam modes vs num_beams=1 at the bench shape (B utterances x 128 text tokens)."""
import sys
import os
import time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import gpt, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_gen = int(sys.argv[2]) if len(sys.argv) > 2 else 120
gcfg = dict(synth.GPT_V25)
m = gpt.UnifiedVoice(spk_cond_mode="campplus", **gcfg, precision="bf16", device="cuda:0")
m.load_state_dict(synth.gpt_weights(gcfg, suppress_eos=True))
g = torch.Generator().manual_seed(0)
text = torch.randint(2, 12000, (B, 128), generator=g).cuda()
langs = torch.full((B,), 3, dtype=torch.long).cuda()
style = torch.randn(1, 192, generator=g).cuda()
emo = (torch.randn(1, 1280, generator=g) * 0.1).cuda()
MODES = ((1, True, False), (3, True, False), (3, False, False), (1, True, True), (3, True, True))
for nb, samp, typ in MODES[: int(os.environ.get("ITTS_BEAM_BENCH_MODES", len(MODES)))]:
    kw = dict(do_sample=samp, top_p=0.8, top_k=30, temperature=0.8, num_beams=nb, repetition_penalty=10.0,
              length_penalty=0.0, typical_sampling=typ, typical_mass=0.9)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        codes, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style,
                                      max_generate_length=n_gen, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"B={B} num_beams={nb} do_sample={samp} typical={typ}: {dt * 1e3:.1f} ms for {codes.shape[1]} tokens "
          f"-> {dt * 1e3 / max(1, codes.shape[1]):.3f} ms/token  timing={m.last_timing}", flush=True)
