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
    g = gpt.UnifiedVoice(layers=2, model_dim=128, heads=2, max_text_tokens=40, max_mel_tokens=60, number_text_tokens=200,
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
