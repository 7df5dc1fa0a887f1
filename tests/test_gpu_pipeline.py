"""GPU test of the IndexTTS2 boundary class on the real HIP engines (small random models) with a stub frontend:
batched multi-segment synthesis must equal segment-by-segment synthesis (what the reference loop does)."""
import numpy as np
import pytest
import torch

from oracle import bigvgan_oracle as BO
from oracle import gpt_oracle as G
from tests.pipeline_stubs import StubFrontend

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build():
    from indextts_amd import bigvgan, gpt
    from indextts_amd.infer_v2_5 import IndexTTS2
    cfg = G.GPTConfig(layers=2, model_dim=128, heads=2, max_text_tokens=40, max_mel_tokens=60, number_text_tokens=200)
    sd = G.synth_weights(cfg, seed=31)
    sd["mel_head.bias"][cfg.stop_mel_token] += 2.0
    g = gpt.UnifiedVoice(spk_cond_mode="campplus", layers=2, model_dim=128, heads=2, max_text_tokens=40, max_mel_tokens=60, number_text_tokens=200,
                         precision="fp32", device=DEV)
    g.load_state_dict(sd)
    h = dict(BO.V2_HPARAMS, upsample_initial_channel=512)
    v = bigvgan.BigVGAN(h)
    v.load_state_dict(BO.synth_weights(h, seed=32))
    v.to(DEV)
    fe = StubFrontend(128, device=DEV)
    return IndexTTS2(cfg={"gpt": {"stop_mel_token": 8193}}, device=DEV, frontend=fe, gpt=g, bigvgan=v)


def test_batched_segments_equal_sequential_segments():
    tts = build()
    kw = dict(num_beams=1, top_k=1, max_mel_tokens=24)          # greedy through infer(): SURVEY.md section 9 item 5
    sr, full = tts.infer("spk.wav", "hello world. a much longer second sentence here. ok", None, "en", **kw)
    parts = [tts.infer("spk.wav", s, None, "en", **kw)[1] for s in ("hello world", "a much longer second sentence here", "ok")]
    sil = np.zeros((int(22050 * 0.2), 1), dtype=np.int16)
    seq = np.concatenate([parts[0], sil, parts[1], sil, parts[2]], axis=0)
    assert sr == 22050 and full.shape == seq.shape
    assert np.abs(full.astype(np.int32) - seq.astype(np.int32)).max() <= 1        # int16 rounding of identical floats


def test_default_generation_mode_runs_beam_sample():
    """No generation kwargs = the reference defaults: do_sample, top_p 0.8, top_k 30, T 0.8, 3 beams, rep-penalty 10."""
    tts = build()
    sr, wav = tts.infer("spk.wav", "default decoding mode. two segments", None, "en", max_mel_tokens=16)
    assert sr == 22050 and wav.shape[0] > 0 and wav.dtype == np.int16
    assert np.abs(wav).max() > 0


# ---- v1 / v1.5 pipeline (indextts/infer.py::IndexTTS): BASELINE configs[0] in miniature -------------------------------
def build_v1():
    from indextts_amd import bigvgan, gpt
    from indextts_amd.infer import IndexTTS
    from tests.pipeline_stubs import StubFrontendV1
    cfg = G.GPTConfig(layers=2, model_dim=128, heads=2, max_text_tokens=40, max_mel_tokens=60, number_text_tokens=200)
    sd = G.synth_weights(cfg, seed=41)
    sd["mel_head.bias"][cfg.stop_mel_token] += 1.8
    g = gpt.UnifiedVoiceV1(layers=2, model_dim=128, heads=2, max_text_tokens=40, max_mel_tokens=60, number_text_tokens=200,
                           precision="fp32", device=DEV)
    g.load_state_dict(sd)
    g.post_init_gpt2_config(kv_cache=True)
    h = dict(BO.V2_HPARAMS, upsample_initial_channel=512, use_tanh_at_final=True, use_bias_at_final=True,
             upsample_rates=[4, 4, 4, 4, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4])
    bsd = BO.synth_weights(h, seed=42, cond_dim=64, in_dim=128, post_gain=0.2)
    proj = torch.randn(100, 64, generator=torch.Generator().manual_seed(43)) * 0.1
    spk_enc = lambda mel_ref, lens=None: (mel_ref.float().mean(dim=1) @ proj.to(mel_ref.device))   # (1,T,100) -> (1,64)
    v = bigvgan.BigVGAN(h, cond_dim=64, in_channels=128, speaker_encoder=spk_enc)
    v.load_state_dict(bsd)
    v.to(DEV)
    fe = StubFrontendV1(128, device=DEV)
    tts = IndexTTS(cfg={"gpt": {"stop_mel_token": 8193, "stop_text_token": 1, "start_text_token": 0}, "version": 1.5},
                   device=DEV, use_fp16=False, frontend=fe, gpt=g, bigvgan=v)
    return tts, cfg, sd, h, bsd, fe, proj


def _v1_oracle_segment(cfg, sd, h, bsd, fe, proj, ids, max_gen):
    gp = G.GenParams(do_sample=False, num_beams=1, repetition_penalty=10.0, max_generate_length=max_gen)
    with torch.no_grad():
        codes = G.inference_speech(sd, cfg, fe.latent, ids, None, gp, kv_cache=True)
        stop = (codes[0] == cfg.stop_mel_token).nonzero()
        n = int(stop[0]) if stop.numel() else codes.shape[1]
        lat = G.forward_latent_v1(sd, cfg, fe.latent, ids, torch.tensor([ids.shape[1]]), codes[:, :n], torch.tensor([n]) * 1024)
    return lat


def test_v1_infer_matches_oracle_composition():
    """IndexTTS.infer (greedy): codes -> remove_long_silence -> latent pass -> speaker-conditioned BigVGAN -> PCM scaling,
    segment by segment, equals the CPU oracle chained the same way."""
    tts, cfg, sd, h, bsd, fe, proj = build_v1()
    text = "alpha beta gamma delta. epsilon zeta eta theta iota kappa."
    sr, wav = tts.infer("prompt.wav", text, None, do_sample=False, num_beams=1, max_mel_tokens=20)
    assert sr == 24000 and wav.dtype == np.int16
    tok = fe.tokenizer
    spk = (fe.mel.transpose(1, 2).mean(dim=1) @ proj)
    ref = []
    for sent in tok.split_segments(tok.tokenize(text), 120):
        ids = torch.tensor(tok.convert_tokens_to_ids(sent), dtype=torch.int32)[None]
        lat = _v1_oracle_segment(cfg, sd, h, bsd, fe, proj, ids, 20)
        with torch.no_grad():
            w = BO.bigvgan_forward(bsd, lat.transpose(1, 2), h, spk=spk.unsqueeze(-1))
        ref.append(torch.clamp(32767 * w.squeeze(1), -32767.0, 32767.0))
    ref = torch.cat(ref, dim=1).type(torch.int16).numpy().T
    assert wav.shape == ref.shape
    assert np.abs(wav.astype(np.int32) - ref.astype(np.int32)).max() <= 2          # int16 truncation of ~1e-6-close floats


def test_v1_infer_fast_buckets_and_chunks():
    """infer_fast: bucketed left-padded decode gives the per-segment codes (padding invariance), latents are B=1 passes,
    the vocoder runs on pairs of concatenated latents (infer.py:459-483)."""
    tts, cfg, sd, h, bsd, fe, proj = build_v1()
    text = "one two three. four five six seven eight nine. ten. eleven twelve thirteen fourteen. fifteen sixteen."
    sr, wav = tts.infer_fast("prompt.wav", text, None, do_sample=False, num_beams=1, max_mel_tokens=16,
                             segments_bucket_max_size=2)
    tok = fe.tokenizer
    spk = (fe.mel.transpose(1, 2).mean(dim=1) @ proj)
    lats = []
    for sent in tok.split_segments(tok.tokenize(text), 100):
        ids = torch.tensor(tok.convert_tokens_to_ids(sent), dtype=torch.int32)[None]
        lats.append(_v1_oracle_segment(cfg, sd, h, bsd, fe, proj, ids, 16))
    ref = []
    for i in range(0, len(lats), 2):
        with torch.no_grad():
            w = BO.bigvgan_forward(bsd, torch.cat(lats[i:i + 2], dim=1).transpose(1, 2), h, spk=spk.unsqueeze(-1))
        ref.append(torch.clamp(32767 * w.squeeze(1), -32767.0, 32767.0))
    ref = torch.cat(ref, dim=1).type(torch.int16).numpy().T
    assert sr == 24000 and wav.shape == ref.shape
    assert np.abs(wav.astype(np.int32) - ref.astype(np.int32)).max() <= 2
    assert fe.calls[0] == ("cond_mel", "prompt.wav", 50)
