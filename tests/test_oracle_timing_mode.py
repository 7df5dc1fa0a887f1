"""bench.py's cpu_baseline leg times the oracles in TIMING_MODE (oracle/bigvgan_oracle.py, oracle/s2mel_oracle.py: the resamplers as strided depthwise
convolutions and the fused CPU attention -- the forms the reference's own modules call, resample.py:29-38,55-58 and gpt_fast/model.py:303), because the
checked forms are up to 2.7 x slower on CPU than the reference (profiles/r03z_cpu/reference_cpu_timing.log) and a baseline must not be slower than
what it stands for.  The timed arithmetic must still be the checked arithmetic: same taps, same padding, same mask -- only the summation order
differs."""
import torch

from oracle import bigvgan_oracle as BO
from oracle import s2mel_oracle as SO


def test_timing_mode_is_off_by_default():
    assert BO.TIMING_MODE is False and SO.TIMING_MODE is False


def test_activation1d_timing_mode_equals_index_form():
    g = torch.Generator().manual_seed(3)
    for B, C, T in ((1, 3, 1), (2, 5, 2), (1, 4, 7), (2, 6, 301)):          # incl. the tiny lengths where every tap hits the replicate padding
        x = torch.randn(B, C, T, generator=g) * 1.5
        al, be = torch.rand(C, generator=g) - 0.5, torch.rand(C, generator=g) - 0.5
        ref = BO.activation1d(x, al, be)
        BO.TIMING_MODE = True
        try:
            got = BO.activation1d(x, al, be)
        finally:
            BO.TIMING_MODE = False
        assert got.shape == ref.shape
        assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))


def test_bigvgan_timing_mode_equals_checked_form():
    h = dict(BO.V2_HPARAMS, upsample_initial_channel=256)
    sd = BO.synth_weights(h, seed=5)
    mel = torch.randn(2, h["num_mels"], 9, generator=torch.Generator().manual_seed(6)) * 2 - 4
    with torch.no_grad():
        ref = BO.bigvgan_forward(sd, mel, h)
        BO.TIMING_MODE = True
        try:
            got = BO.bigvgan_forward(sd, mel, h)
        finally:
            BO.TIMING_MODE = False
    assert float((got - ref).abs().max()) <= 1e-5


def test_s2mel_timing_mode_equals_checked_form_with_padded_keys():
    cfg = SO.S2MelConfig(hidden_dim=64, num_heads=2, depth=5, in_channels=80, content_dim=48, style_dim=24, wavenet_hidden=64, wavenet_layers=3,
                         wavenet_kernel=5, wavenet_dilation_rate=2)
    sd = SO.synth_weights(cfg, 61)
    g = torch.Generator().manual_seed(62)
    T, Tp = 57, 19
    z = torch.randn(1, cfg.in_channels, T, generator=g)
    prompt = torch.randn(1, cfg.in_channels, Tp, generator=g) * 0.5 - 1.0
    mu = torch.randn(1, T, cfg.content_dim, generator=g)
    style = torch.randn(1, cfg.style_dim, generator=g)
    lens = torch.tensor([T - 6])                                           # six padded frames: the key mask bites
    with torch.no_grad():
        ref = SO.cfm_solve_euler(sd, cfg, z, lens, prompt, mu, style, 4, 0.7)
        SO.TIMING_MODE = True
        try:
            got = SO.cfm_solve_euler(sd, cfg, z, lens, prompt, mu, style, 4, 0.7)
        finally:
            SO.TIMING_MODE = False
    assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
