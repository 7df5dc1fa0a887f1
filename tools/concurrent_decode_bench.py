"""GPU box: does splitting the 64-row decode into concurrent half / quarter batches (own stream, own engine handle each) hide the
launch-latency gaps of the decode step?  Prints wall ms per token-step of the whole 64-row job for 1 x 64, 2 x 32, 4 x 16."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import gpt, synth  # noqa: E402

B = 64
n_gen = int(sys.argv[1]) if len(sys.argv) > 1 else 280
gcfg = dict(synth.GPT_V25)
sd = synth.gpt_weights(gcfg, suppress_eos=True)
g = torch.Generator().manual_seed(0)
text = torch.randint(2, 12000, (B, 128), generator=g).cuda()
langs = torch.full((B,), 3, dtype=torch.long).cuda()
style = torch.randn(1, 192, generator=g).cuda()
emo = (torch.randn(1, 1280, generator=g) * 0.1).cuda()
kw = dict(do_sample=True, top_p=0.8, top_k=30, temperature=0.8, num_beams=1, repetition_penalty=10.0, length_penalty=0.0)
models = []
for parts in (1, 2, 4):
    while len(models) < parts:
        m = gpt.UnifiedVoice(spk_cond_mode="campplus", **gcfg, precision="bf16", device="cuda:0")
        m.load_state_dict(sd)
        models.append(m)
    streams = [torch.cuda.Stream() for _ in range(parts)]
    rows = B // parts

    def work(i):
        with torch.cuda.stream(streams[i]):
            sl = slice(i * rows, (i + 1) * rows)
            models[i].inference_speech(None, text[sl], langs=langs[sl], emo_vec=emo, campplus_embedding=style, max_generate_length=n_gen, **kw)
            streams[i].synchronize()

    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(parts)]
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{parts} x {rows} rows concurrently: {dt * 1e3:.1f} ms for {n_gen} tokens -> {dt * 1e3 / n_gen:.3f} ms per 64-row token step "
          f"(timing of part 0: {models[0].last_timing})", flush=True)
