#!/usr/bin/env python3
"""Progressive timing of the hot path at growing sizes (debug aid for the GPU box; every line is flushed)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indextts_amd import bigvgan, gpt, synth

def log(*a):
    print(f"[{time.time() - T0:7.2f}s]", *a, flush=True)

T0 = time.time()
dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("gpt", "both"):
    gcfg = dict(synth.GPT_V25)
    gsd = synth.gpt_weights(gcfg, seed=1234, suppress_eos=True)
    log("gpt weights synthesised")
    m = gpt.UnifiedVoice(spk_cond_mode="campplus", **gcfg, precision="bf16", device=dev)
    m.load_state_dict(gsd)
    log("gpt loaded")
    style = torch.randn(1, 192).to(dev); emo = (torch.randn(1, 1280) * 0.1).to(dev)
    for B, n_gen, graph in [(2, 16, False), (2, 16, True), (8, 64, True), (64, 64, True), (64, 64, False), (64, 560, True)]:
        text = torch.randint(2, 12000, (B, 128)).to(dev)
        langs = torch.full((B,), 3, dtype=torch.long, device=dev)
        m.use_graph = graph
        t = time.time()
        codes, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=n_gen,
                                      do_sample=True, top_p=0.8, top_k=30, temperature=0.8, num_beams=1, repetition_penalty=10.0)
        torch.cuda.synchronize()
        log(f"gpt B={B} n={n_gen} graph={graph}: {time.time() - t:.3f}s codes {tuple(codes.shape)} timing {m.last_timing}")
if which in ("voc", "both"):
    bh = dict(synth.BIGVGAN_V2_22K)
    voc = bigvgan.BigVGAN(bh)
    voc.load_state_dict(synth.bigvgan_weights(bh))
    voc.to(dev)
    voc.set_profiling(True)
    log("bigvgan loaded")
    for B, T in [(1, 100), (2, 400), (8, 1926), (16, 1926), (64, 1926)]:
        mel = (torch.randn(B, 80, T) * 2 - 4).to(dev)
        for rep in range(2):
            t = time.time()
            w = voc(mel)
            torch.cuda.synchronize()
            dt = time.time() - t
            p = voc.profile()
            log(f"bigvgan B={B} T={T} rep{rep}: {dt:.3f}s  " + " ".join(f"{k}:{v['ms']:.1f}ms/{v['launches']}" for k, v in p.items())
                + f" conv TF={p['conv1d_mfma']['flops'] / max(p['conv1d_mfma']['ms'], 1e-9) / 1e9:.1f}")
log("done")
