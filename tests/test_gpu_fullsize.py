"""GPU parity at the shapes `bench.py` times: every stage of the headline against the CPU oracle / the reference's
own ids at BASELINE.json configs[2]'s sizes, not at miniature fixtures.

  * BigVGAN: full width (1536 channels), ragged rows of 1926 and 1409 mel frames, waveform RMS <= 1e-4 vs `oracle/bigvgan_oracle.py`
    (indextts/s2mel/modules/bigvgan/bigvgan.py:360-386) -- multi-tile, XCD-mapped, 24-co-tile conv launches;
  * CFM: production configuration (DiT 13 x 512 x 8 heads, WaveNet 8 x 512), two utterances with a 517-frame prompt and 1926 / 1900
    target frames, three CFG Euler steps, f32 engine mode <= 1e-4 vs `oracle/s2mel_oracle.py` (flow_matching.py:57-115);
  * the bf16 s2mel mode after the FULL 25 steps at that size: mel error and the error of the waveform BigVGAN makes of it, against the
    f32 engine mode (the reference runs this stage in fp32, infer_v2_5.py:827-828) -- the numbers that decide which precision may
    carry the headline (north_star: waveform within 1e-4 RMS);
  * GPT: f32 engine ids == the ids the REFERENCE's own classes produced at full size out to context 694 (128 text tokens + 560
    generated; tests/golden/gpt_fullsize_ctx694.npz, tools/make_golden_gpt.py fullsize).
CPU oracle cost on the GPU box (16 threads): BigVGAN ~16 s, CFM ~30 s.
"""
import os

import numpy as np
import pytest
import torch

from oracle import bigvgan_oracle as BO
from oracle import gpt_oracle as G
from oracle import s2mel_oracle as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
WAVE_RMS_TOL = 1e-4                      # north_star: BigVGAN waveform within 1e-4 RMS of the reference CPU path
CFM_F32_TOL = 1e-4                       # absolute, on mel values of RMS ~1 (tests/test_gpu_s2mel.py::F32_TOL)


def rms(a):
    return float(torch.as_tensor(a).double().pow(2).mean().sqrt())


def _threads():
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(n, 32)))


# ---- BigVGAN -------------------------------------------------------------------------------------------------------------------
def test_bigvgan_full_width_bench_length_vs_oracle():
    from indextts_amd import bigvgan
    _threads()
    h = dict(BO.V2_HPARAMS)
    assert h["upsample_initial_channel"] == 1536
    sd = BO.synth_weights(h, seed=1234)
    m = bigvgan.BigVGAN(h)
    m.load_state_dict(sd)
    m.to(DEV)
    g = torch.Generator().manual_seed(77)
    lens = [1926, 1409]                                            # int(2 * 560 * 1.72) and a ragged shorter row
    mel = torch.randn(2, 80, max(lens), generator=g) * 2 - 4
    mel[1, :, lens[1]:] = 0
    wav = m(mel.to(DEV), lens=torch.tensor(lens, dtype=torch.int32)).cpu()
    assert wav.shape == (2, 1, max(lens) * 256)
    for b, n in enumerate(lens):
        with torch.no_grad():
            ref = BO.bigvgan_forward(sd, mel[b:b + 1, :, :n], h)
        e, sig = rms(wav[b:b + 1, :, : n * 256] - ref), rms(ref)
        print(f"BigVGAN 1536 ch x {n} frames (row {b} of a ragged batch): waveform rms error vs oracle {e:.3e} (signal rms {sig:.3f}, "
              f"relative {e / sig:.2e})")
        assert sig > 0.05 and e <= WAVE_RMS_TOL


# ---- CFM -----------------------------------------------------------------------------------------------------------------------
def _cfm_case():
    cfg = S.S2MelConfig()
    assert (cfg.depth, cfg.hidden_dim, cfg.wavenet_layers, cfg.wavenet_hidden) == (13, 512, 8, 512)
    sd = S.synth_weights(cfg, 3)
    g = torch.Generator().manual_seed(11)
    Tp, tgt = 517, [1926, 1900]
    T = [Tp + t for t in tgt]
    Tm = max(T)
    x = torch.randn(2, 80, Tm, generator=g)
    mu = torch.randn(2, Tm, cfg.content_dim, generator=g)
    prompt = torch.randn(1, 80, Tp, generator=g) * 2 - 4
    style = torch.randn(1, cfg.style_dim, generator=g)
    return cfg, sd, x, mu, prompt, style, Tp, T


def _cfm_engine(cfg, sd, precision):
    from tests.test_gpu_s2mel import engine
    return engine(cfg, sd, precision)


def test_cfm_production_config_bench_frames_f32_vs_oracle():
    _threads()
    cfg, sd, x, mu, prompt, style, Tp, T = _cfm_case()
    m = _cfm_engine(cfg, sd, "fp32")
    n_steps = 3
    y = m.solve_euler(x.clone(), torch.tensor(T), prompt, mu, style, None, torch.linspace(0, 1, n_steps + 1), 0.7, frame_lens=T).cpu()
    for u in range(2):
        with torch.no_grad():
            ref = S.cfm_solve_euler(sd, cfg, x[u:u + 1, :, : T[u]], torch.tensor([T[u]]), prompt, mu[u:u + 1, : T[u]], style, n_steps, 0.7)
        d = y[u:u + 1, :, : T[u]] - ref
        print(f"CFM 13 x 512 / WaveNet 8 x 512, {Tp} + {T[u] - Tp} frames, {n_steps} CFG Euler steps (utt {u} of a packed pair): "
              f"f32 engine vs oracle max|d| {float(d.abs().max()):.3e}, rms {rms(d):.3e} (output rms {rms(ref):.3f})")
        assert float(d.abs().max()) <= CFM_F32_TOL
        assert float(y[u, :, :Tp].abs().max()) == 0.0


def test_cfm_modes_error_after_25_steps_mel_and_waveform():
    """What the other s2mel modes cost at the benchmarked size after the full 25-step solve: mel RMS error and the RMS error of the
    BigVGAN waveform, against the native-f32 engine mode (pinned to the oracle at this size by the test above).  north_star's bar
    for the waveform is 1e-4 RMS: the measured figures decide which mode may carry `bench.py`'s headline.
      bf16    bf16 GEMM operands / Q, K, V / probabilities: reported; far above the bar -> never the headline;
      fp32x3  f32 activations, every GEMM operand carried exactly as three bf16 planes (gemm_x3_kernel), attention / norms / gates in
              f32: must sit at rounding level, orders of magnitude under the bar."""
    from indextts_amd import bigvgan
    cfg, sd, x, mu, prompt, style, Tp, T = _cfm_case()
    t_span = torch.linspace(0, 1, 26)
    mels = {}
    for prec in ("fp32", "fp32x3", "bf16"):
        m = _cfm_engine(cfg, sd, prec)
        mels[prec] = m.solve_euler(x.clone(), torch.tensor(T), prompt, mu, style, None, t_span, 0.7, frame_lens=T)[:, :, Tp:].contiguous()
        del m
        torch.cuda.empty_cache()
    lens = torch.tensor([t - Tp for t in T], dtype=torch.int32)
    h = dict(BO.V2_HPARAMS)
    v = bigvgan.BigVGAN(h)
    v.load_state_dict(BO.synth_weights(h, seed=1234))
    v.to(DEV)
    wav = {k: v(mv, lens=lens).cpu() for k, mv in mels.items()}
    for prec in ("fp32x3", "bf16"):
        for u in range(2):
            n = int(lens[u])
            dm = (mels[prec][u, :, :n] - mels["fp32"][u, :, :n]).cpu()
            dw = wav[prec][u, :, : n * 256] - wav["fp32"][u, :, : n * 256]
            mel_rms, wav_rms, ref_rms = rms(dm), rms(dw), rms(mels["fp32"][u, :, :n])
            print(f"{prec} s2mel after 25 steps at {Tp} + {n} frames (utt {u}): mel rms error {mel_rms:.3e} (mel rms {ref_rms:.3f}, relative "
                  f"{mel_rms / ref_rms:.2e}); waveform rms error {wav_rms:.3e} (waveform rms {rms(wav['fp32'][u, :, : n * 256]):.3f}) -> "
                  f"{'within' if wav_rms <= WAVE_RMS_TOL else 'ABOVE'} north_star's 1e-4 waveform bar")
            assert np.isfinite(mel_rms) and np.isfinite(wav_rms)
            if prec == "fp32x3":
                assert wav_rms <= WAVE_RMS_TOL / 10 and mel_rms <= 1e-4          # rounding level: an order under the bar at least
            else:
                assert mel_rms / ref_rms <= 0.10                                 # sanity bound; the bf16 mode is reported, never the parity mode


# ---- GPT -----------------------------------------------------------------------------------------------------------------------
def test_gpt_f32_ids_vs_reference_context_694(golden_dir):
    from indextts_amd import gpt
    z = np.load(os.path.join(golden_dir, "gpt_fullsize_ctx694.npz"))
    c = [int(v) for v in z["cfg"]]
    cfg = G.GPTConfig(layers=c[0], model_dim=c[1], heads=c[2], max_text_tokens=c[3], max_mel_tokens=c[4], number_text_tokens=c[5])
    assert (cfg.layers, cfg.model_dim, cfg.heads) == (24, 1280, 20)
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] -= 1e4
    n = int(z["n"])
    m = gpt.UnifiedVoice(spk_cond_mode="campplus", layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads,
                         max_text_tokens=cfg.max_text_tokens, max_mel_tokens=cfg.max_mel_tokens, number_text_tokens=cfg.number_text_tokens,
                         precision="fp32", device=DEV)
    m.load_state_dict(sd)
    m.post_init_gpt2_config(kv_cache=True)
    ids, _ = m.inference_speech(None, torch.from_numpy(z["text"]), langs=torch.from_numpy(z["langs"]), emo_vec=torch.from_numpy(z["emo_vec"]),
                                campplus_embedding=torch.from_numpy(z["style"]), max_generate_length=n, do_sample=False, num_beams=1,
                                repetition_penalty=10.0)
    got, ref, margins = ids.cpu().numpy(), z["codes"], z["margins"]
    assert got.shape == ref.shape == (2, n)
    ctx = 3 + int(z["lens"][0]) + 2 + 1 + n
    print(f"GPT 24 x 1280 f32 engine vs the reference's ids: {n} greedy steps, context up to {ctx}; min fp32 top-2 margin of the "
          f"processed scores over the run {float(margins.min()):.3e}")
    if not np.array_equal(got, ref):
        r, s = np.argwhere(got != ref)[0]
        pytest.fail(f"row {r} step {s}: engine {got[r, s]} reference {ref[r, s]} (fp32 margin there {margins[r, s]:.2e})")


def test_gpt_f32_beam_sample_ids_vs_reference_full_size(golden_dir):
    """The reference's DEFAULT generation mode (3-beam beam-sample: do_sample, top_p 0.8, top_k 30, temperature 0.8, repetition_penalty 10,
    length_penalty 0 -- infer_v2_5.py:732-740) on the full-size stack (24 x 1280 x 20 heads), two utterances of 64 / 47 text tokens, 200 steps:
    the f32 engine's ids equal the ids the REFERENCE's own classes produced (vendored GenerationMixin._beam_search + BeamSearchScorer over HF
    GPT2Model, run on CPU by tools/make_golden_gpt.py fullsize_beam with the same explicit uniform stream; tests/golden/gpt_fullsize_beam3.npz).
    (the beam fixtures were small-model only.)"""
    from indextts_amd import gpt
    z = np.load(os.path.join(golden_dir, "gpt_fullsize_beam3.npz"))
    c = [int(v) for v in z["cfg"]]
    cfg = G.GPTConfig(layers=c[0], model_dim=c[1], heads=c[2], max_text_tokens=c[3], max_mel_tokens=c[4], number_text_tokens=c[5])
    assert (cfg.layers, cfg.model_dim, cfg.heads) == (24, 1280, 20)
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] -= 1e4
    n, g = int(z["n"]), z["gen"]
    m = gpt.UnifiedVoice(spk_cond_mode="campplus", layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads,
                         max_text_tokens=cfg.max_text_tokens, max_mel_tokens=cfg.max_mel_tokens, number_text_tokens=cfg.number_text_tokens,
                         precision="fp32", device=DEV)
    m.load_state_dict(sd)
    m.post_init_gpt2_config(kv_cache=True)
    ids, _ = m.inference_speech(None, torch.from_numpy(z["text"]), langs=torch.from_numpy(z["langs"]), emo_vec=torch.from_numpy(z["emo_vec"]),
                                campplus_embedding=torch.from_numpy(z["style"]), max_generate_length=n, uniforms=torch.from_numpy(z["uniforms"]),
                                do_sample=bool(g[0]), num_beams=int(g[1]), top_p=float(g[2]), top_k=int(g[3]), temperature=float(g[4]),
                                repetition_penalty=float(g[5]), length_penalty=float(g[6]))
    got, ref = ids.cpu().numpy(), z["codes"]
    print(f"GPT 24 x 1280 f32 engine, 3-beam beam-sample vs the reference's ids: {ref.shape[1]} codes x {ref.shape[0]} utterances")
    assert got.shape == ref.shape
    if not np.array_equal(got, ref):
        r, s = np.argwhere(got != ref)[0]
        pytest.fail(f"row {r} step {s}: engine {got[r, s]} reference {ref[r, s]} ({int((got != ref).sum())} ids differ)")
