#!/usr/bin/env python3
"""How representative is bench.py's `cpu_baseline` (kind "port": the oracle) of the REFERENCE's own CPU path?

Runs only in the build container (needs /root/reference; nothing here travels to the GPU box).  The reference's own classes --
`UnifiedVoice` / `GPT2InferenceModel` over HF GPT2Model (indextts/gpt/model_v2.py), `CFM` + `DiT` (indextts/s2mel/modules/flow_matching.py,
diffusion_transformer.py) and `BigVGAN` (indextts/s2mel/modules/bigvgan/bigvgan.py) -- are built at the production widths with the seeded synthetic
weights and timed on this container's cores on the same bounded samples `bench.py::cpu_baseline` times the oracle on; the oracle is timed beside them
in the same process.  Output: one line per stage with both costs and their ratio, and the audio-seconds/second both extrapolate to for the benchmark's
utterance (560 tokens / 964 generated frames behind a 481-frame prompt / 25 CFG Euler steps).  The log is committed as profiles/r03z_cpu/.

usage: reference_cpu_timing.py [threads]           (about 3 minutes on 8 threads)"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
import make_golden_gpt as MG  # noqa: E402  (imports transformers: before the librosa / torchaudio stubs below, which its availability probes trip over)
import make_golden_bigvgan as MB  # noqa: E402  (stubs librosa with the two names meldataset.py imports, then imports the reference vocoder)
import ref_shim_s2mel as R  # noqa: E402

R.install()
import make_golden_s2mel as MS  # noqa: E402
from oracle import bigvgan_oracle as BO  # noqa: E402
from oracle import gpt_oracle as GO  # noqa: E402
from oracle import s2mel_oracle as SO  # noqa: E402

HOP, SR, EULER = 256, 22050, 25


def best(f, n=2):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def gpt_leg(n_text=128, steps=120):
    cfg = GO.GPTConfig(max_text_tokens=140, max_mel_tokens=600)
    sd = GO.synth_weights(cfg, seed=1234)
    sd["mel_head.bias"][cfg.stop_mel_token] -= 1e4
    g = torch.Generator().manual_seed(7)
    text = torch.randint(2, cfg.number_text_tokens, (1, n_text), generator=g)
    style = torch.randn(1, 192, generator=g)
    emo = torch.randn(1, cfg.model_dim, generator=g) * 0.1
    langs = torch.zeros(1, dtype=torch.long)
    gk = dict(do_sample=False, num_beams=1, repetition_penalty=10.0)
    uv = MG.build_reference(sd, cfg, kv_cache=True)
    out = {}
    with torch.no_grad():
        for n in (8, 8 + steps):            # two lengths: the difference is `steps` decode steps, prefill and set-up cancel
            t0 = time.perf_counter()
            codes_r, _ = uv.inference_speech(torch.zeros(1, 4, 2), text, langs=langs, emo_vec=emo, campplus_embedding=style,
                                             max_generate_length=n, **gk)
            out[("ref", n)] = time.perf_counter() - t0
            t0 = time.perf_counter()
            codes_o = GO.inference_speech(sd, cfg, GO.conds_latent_campplus(sd, style, emo), text, langs,
                                          GO.GenParams(max_generate_length=n, **gk), kv_cache=True)
            out[("oracle", n)] = time.perf_counter() - t0
            assert codes_r.shape == codes_o.shape and bool((codes_r == codes_o).all()), "oracle ids differ from the reference's"
    ref = (out[("ref", 8 + steps)] - out[("ref", 8)]) / steps
    ora = (out[("oracle", 8 + steps)] - out[("oracle", 8)]) / steps
    return ref, ora, out[("ref", 8)], out[("oracle", 8)]


def s2mel_leg(T=1445, Tp=481):
    cfg = SO.S2MelConfig()
    sd = SO.synth_weights(cfg, 3)
    m = MS.reference(cfg, sd)
    m.estimator.setup_caches(max_batch_size=2, max_seq_length=max(2048, T))
    g = torch.Generator().manual_seed(5)
    z = torch.randn(1, cfg.in_channels, T, generator=g)
    prompt = torch.randn(1, cfg.in_channels, Tp, generator=g)
    mu = torch.randn(1, T, cfg.content_dim, generator=g)
    style = torch.randn(1, cfg.style_dim, generator=g)
    lens = torch.tensor([T])
    res = {}
    with torch.no_grad():
        y_r = m.solve_euler(z.clone(), lens, prompt, mu.clone(), style, None, torch.linspace(0, 1, 2), inference_cfg_rate=0.7)
        y_o = SO.cfm_solve_euler(sd, cfg, z, lens, prompt, mu, style, 1, 0.7)
        res["diff"] = float((y_r - y_o).abs().max())
        res["ref"] = best(lambda: m.solve_euler(z.clone(), lens, prompt, mu.clone(), style, None, torch.linspace(0, 1, 2), inference_cfg_rate=0.7))
        res["oracle"] = best(lambda: SO.cfm_solve_euler(sd, cfg, z, lens, prompt, mu, style, 1, 0.7))
        SO.TIMING_MODE = True
        try:
            res["timing_diff"] = float((SO.cfm_solve_euler(sd, cfg, z, lens, prompt, mu, style, 1, 0.7) - y_o).abs().max())
            res["oracle_timing"] = best(lambda: SO.cfm_solve_euler(sd, cfg, z, lens, prompt, mu, style, 1, 0.7))
        finally:
            SO.TIMING_MODE = False
    return res


def bigvgan_leg(frames=480):
    h = dict(BO.V2_HPARAMS)
    sd = BO.synth_weights(h, seed=1234)
    model = MB.ref_model(h, sd)
    g = torch.Generator().manual_seed(9)
    mel = torch.randn(1, h["num_mels"], frames, generator=g) * 2 - 4
    with torch.no_grad():
        w_r = model(mel[:, :, :16])
        w_o = BO.bigvgan_forward(sd, mel[:, :, :16], h)
        d = float((w_r - w_o).abs().max())
        ref = best(lambda: model(mel)) / frames
        ora = best(lambda: BO.bigvgan_forward(sd, mel, h)) / frames
        BO.TIMING_MODE = True
        try:
            dt = float((BO.bigvgan_forward(sd, mel[:, :, :16], h) - w_o).abs().max())
            orat = best(lambda: BO.bigvgan_forward(sd, mel, h)) / frames
        finally:
            BO.TIMING_MODE = False
    return ref, ora, d, orat, dt


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    print(f"reference classes vs oracle on {threads} CPU threads of the build container (torch {torch.__version__}, fp32)", flush=True)
    g_ref, g_ora, g_ref0, g_ora0 = gpt_leg()
    print(f"GPT decode, 24 x 1280, 128-token text, greedy, kv-cache: reference {g_ref * 1e3:.1f} ms/token, oracle {g_ora * 1e3:.1f} ms/token "
          f"(oracle / reference {g_ora / g_ref:.2f}); 8-token call incl. prefill and set-up: reference {g_ref0:.2f} s, oracle {g_ora0:.2f} s; "
          f"ids identical", flush=True)
    s = s2mel_leg()
    print(f"s2mel, one CFG Euler step at 1445 frames (DiT 13 x 512 + WaveNet 8 x 512): reference {s['ref']:.2f} s, oracle {s['oracle']:.2f} s "
          f"(oracle / reference {s['oracle'] / s['ref']:.2f}), oracle in TIMING_MODE (fused CPU attention, what cpu_baseline times) {s['oracle_timing']:.2f} s "
          f"({s['oracle_timing'] / s['ref']:.2f}); max|d| of the step's output: oracle vs reference {s['diff']:.2e}, TIMING_MODE vs checked form "
          f"{s['timing_diff']:.2e}", flush=True)
    b_ref, b_ora, b_d, b_orat, b_dt = bigvgan_leg()
    print(f"BigVGAN 1536 ch, 480 frames: reference {b_ref * 1e3:.2f} ms/frame, oracle {b_ora * 1e3:.2f} ms/frame "
          f"(oracle / reference {b_ora / b_ref:.2f}), oracle in TIMING_MODE (resamplers as strided depthwise convolutions) {b_orat * 1e3:.2f} ms/frame "
          f"({b_orat / b_ref:.2f}); max|d| on 16 frames: oracle vs reference {b_d:.2e}, TIMING_MODE vs checked form {b_dt:.2e}", flush=True)
    n_gen, t_mel = 560, 964
    audio = t_mel * HOP / SR
    for name, tok, eul, fr, pre in (("reference", g_ref, s["ref"], b_ref, g_ref0), ("oracle, checked form", g_ora, s["oracle"], b_ora, g_ora0),
                                    ("oracle, TIMING_MODE", g_ora, s["oracle_timing"], b_orat, g_ora0)):
        total = pre + n_gen * tok + EULER * eul + t_mel * fr
        print(f"{name:20s}: {n_gen} tokens {n_gen * tok:.1f} s + {EULER} Euler steps {EULER * eul:.1f} s + {t_mel} frames {t_mel * fr:.1f} s "
              f"(+ {pre:.1f} s prefill / set-up) = {total:.1f} s for {audio:.2f} s of audio -> {audio / total:.4f} audio-seconds/second "
              f"(RTF {total / audio:.1f})", flush=True)


if __name__ == "__main__":
    main()
