"""GPU edge cases of the hot path (the shapes the reference's own code paths special-case): single frames and zero-length
rows in the vocoder, replicate padding on tiny rows, one-token generation, empty / single-token text rows, and the
position-table boundary of the decoder.  Expected values come from the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import bigvgan_oracle as O
from oracle import gpt_oracle as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rms(x):
    return float(x.detach().double().cpu().pow(2).mean().sqrt())


@pytest.fixture(scope="module")
def voc():
    from indextts_amd import bigvgan
    h = dict(O.V2_HPARAMS, upsample_initial_channel=512)
    sd = O.synth_weights(h, seed=77)
    m = bigvgan.BigVGAN(h)
    m.load_state_dict(sd)
    return m.to(DEV), h, sd


@pytest.mark.parametrize("T", [1, 2, 3, 7])
def test_vocoder_tiny_mel(voc, T):
    """1..7 mel frames: every conv/activation window is dominated by zero / replicate padding."""
    m, h, sd = voc
    mel = torch.randn(1, 80, T, generator=torch.Generator().manual_seed(T)) * 2 - 4
    with torch.no_grad():
        ref = O.bigvgan_forward(sd, mel, h)
    wav = m(mel.to(DEV)).cpu()
    assert wav.shape == ref.shape == (1, 1, T * 256)
    assert rms(wav - ref) <= 1e-4


def test_vocoder_zero_length_rows_and_empty_batch(voc):
    m, h, sd = voc
    mel = torch.randn(3, 80, 9, generator=torch.Generator().manual_seed(1)) * 2 - 4
    lens = [9, 0, 4]
    wav = m(mel.to(DEV), lens=lens).cpu()
    with torch.no_grad():
        for b, n in enumerate(lens):
            if n == 0:
                assert float(wav[b].abs().max()) == 0.0
            else:
                ref = O.bigvgan_forward(sd, mel[b:b + 1, :, :n], h)
                assert rms(wav[b:b + 1, :, : n * 256] - ref) <= 1e-4
                if n * 256 < wav.shape[-1]:
                    assert float(wav[b, :, n * 256:].abs().max()) == 0.0
    assert m(torch.zeros(0, 80, 5, device=DEV)).shape == (0, 1, 1280)
    assert m(torch.zeros(2, 80, 0, device=DEV)).shape == (2, 1, 0)


@pytest.mark.parametrize("T", [1, 2, 5, 6, 13])
def test_activation_rows_shorter_than_the_filter(T):
    """Activation1d on rows shorter than the 12-tap filters: replicate padding reaches past both ends at once."""
    from indextts_amd import bigvgan
    g = torch.Generator().manual_seed(T)
    x = torch.randn(2, 3, T, generator=g) * 2
    al, be = torch.rand(3, generator=g) - 0.5, torch.rand(3, generator=g) - 0.5
    f = O.default_filter()
    y = bigvgan.anti_alias_activation(x.to(DEV), f, f, al, be).cpu()
    assert (y - O.activation1d(x, al, be)).abs().max() < 2e-5


def _gpt(cfg, sd):
    from indextts_amd import gpt
    m = gpt.UnifiedVoice(spk_cond_mode="campplus", layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, max_text_tokens=cfg.max_text_tokens,
                         max_mel_tokens=cfg.max_mel_tokens, number_text_tokens=cfg.number_text_tokens, precision="fp32",
                         device=DEV)
    m.load_state_dict(sd)
    return m


def _both(cfg, sd, text, langs, max_gen, kv=True, eos_bias=0.0):
    g = torch.Generator().manual_seed(11)
    style = torch.randn(1, 192, generator=g)
    emo = torch.randn(1, cfg.model_dim, generator=g) * 0.1
    gp = G.GenParams(do_sample=False, num_beams=1, repetition_penalty=10.0, max_generate_length=max_gen)
    conds = G.conds_latent_campplus(sd, style, emo)
    with torch.no_grad():
        ref = G.inference_speech(sd, cfg, conds, text, langs, gp, kv_cache=kv).numpy()
    m = _gpt(cfg, sd)
    m.post_init_gpt2_config(kv_cache=kv)
    codes, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=max_gen,
                                  do_sample=False, num_beams=1, repetition_penalty=10.0)
    return codes.cpu().numpy(), ref


def test_single_new_token_and_single_text_token():
    cfg = G.GPTConfig(layers=2, model_dim=128, heads=2, max_text_tokens=20, max_mel_tokens=30, number_text_tokens=60)
    sd = G.synth_weights(cfg, seed=51)
    text = torch.tensor([[7]])
    got, ref = _both(cfg, sd, text, torch.tensor([2]), 1)
    assert got.shape == (1, 1) and np.array_equal(got, ref)
    got, ref = _both(cfg, sd, text, torch.tensor([2]), 6)
    assert np.array_equal(got, ref)


def test_row_with_no_text_tokens_in_a_batch():
    """A row made only of pad/stop ids is stripped to [start, stop] and fully left-padded (model_v2.py:674-699)."""
    cfg = G.GPTConfig(layers=2, model_dim=128, heads=2, max_text_tokens=20, max_mel_tokens=30, number_text_tokens=60)
    sd = G.synth_weights(cfg, seed=52)
    text = torch.tensor([[5, 9, 33, 2, 17], [1, 1, 1, 1, 1], [44, 1, 1, 1, 1]])
    got, ref = _both(cfg, sd, text, torch.tensor([1, 2, 3]), 8)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("kv", [True, False])
def test_position_table_boundary(kv):
    """Longest text the table allows and max_mel_tokens - 1 generated tokens: the last step reads the last row of the mel
    position table under the kv-cache rule (k+1) and stays inside it under the no-cache rule."""
    cfg = G.GPTConfig(layers=1, model_dim=128, heads=2, max_text_tokens=12, max_mel_tokens=14, number_text_tokens=60)
    sd = G.synth_weights(cfg, seed=53)
    sd["mel_head.bias"][cfg.stop_mel_token] -= 50.0                   # never stop: run to the limit
    text = torch.randint(2, 60, (2, 12), generator=torch.Generator().manual_seed(3))
    got, ref = _both(cfg, sd, text, torch.tensor([0, 1]), cfg.max_mel_tokens - 1, kv=kv)
    assert got.shape == ref.shape == (2, cfg.max_mel_tokens - 1)
    assert np.array_equal(got, ref)


def test_generation_past_the_position_table_is_an_error():
    from indextts_amd import _lib
    cfg = G.GPTConfig(layers=1, model_dim=128, heads=2, max_text_tokens=12, max_mel_tokens=14, number_text_tokens=60)
    m = _gpt(cfg, G.synth_weights(cfg, seed=54))
    with pytest.raises(_lib.HipEngineError):
        m.inference_speech(None, torch.randint(2, 60, (1, 5)), emo_vec=torch.zeros(1, 128), campplus_embedding=torch.zeros(1, 192),
                           max_generate_length=40, do_sample=False, num_beams=1)
