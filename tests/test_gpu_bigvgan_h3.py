"""GPU tests of the opt-in f16 x 3 split-operand conv mode of the vocoder (`BigVGAN(conv_mode="f16x3")`, `itts_conv1d_h3_forward`):
f32 operands split into two f16 parts each (22 significand bits), three exact f16 MFMA products per f32 product.  Unit op against
an f64 torch conv at f32-level tolerance (5e-6 of the output scale: tighter than the 2e-5 the exact-f32 kernel is held to); the
generator against the waveforms the REFERENCE BigVGAN class produced, held to 1e-5 RMS (10x tighter than north_star's 1e-4),
including the `loud` fixture."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import bigvgan_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rms(x):
    return float(np.sqrt(np.mean(np.square(np.asarray(x, dtype=np.float64)))))


H3_CASES = [
    (64, 64, 3, 1, 130, 2),
    (192, 192, 11, 3, 260, 1),     # 1.5 co tiles of 128
    (96, 96, 7, 5, 515, 2),        # 4+ frame tiles, dilated taps reaching 15 frames out
    (32, 48, 3, 1, 77, 3),         # fewer n-tiles than a block stages (clamped), one K tile per tap
    (384, 384, 7, 1, 128, 1),      # exactly one frame tile of the 128-frame kernel
    (64, 64, 11, 5, 300, 2),       # taps reach 25 frames out: beyond the window kernel's 48-frame span -> the 128-frame kernel
    (64, 64, 1, 1, 100, 1),        # k = 1 -> the 128-frame kernel
    (128, 128, 3, 3, 1000, 1),     # four 256-frame tiles of the window kernel, ragged last one
]


@pytest.mark.parametrize("Cin,Cout,k,d,T,B", H3_CASES)
def test_conv1d_h3_vs_f64(Cin, Cout, k, d, T, B):
    from indextts_amd import bigvgan as bv
    g = torch.Generator().manual_seed(Cin * 1000 + Cout + k)
    x = torch.randn(B, Cin, T, generator=g) * 3
    w = torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv1d(x.double(), w.double(), bias.double(), dilation=d, padding=(k - 1) // 2 * d)
    w3 = bv.pack_conv1d_h3_weight(w).to(DEV)
    y = bv.conv1d_h3(x.to(DEV), w3, bias.to(DEV), Cout, k, d).cpu()
    scale = max(1.0, float(ref.abs().max()))
    e32 = float((F.conv1d(x, w, bias, dilation=d, padding=(k - 1) // 2 * d).double() - ref).abs().max())
    err = float((y.double() - ref).abs().max())
    print(f"h3 conv {Cin}x{Cout} k{k} d{d}: max|d| vs f64 {err:.2e} (torch f32 conv: {e32:.2e}, scale {scale:.1f})")
    assert err < 5e-6 * scale
    res = torch.randn(B, Cout, T, generator=g)
    y0 = torch.randn(B, Cout, T, generator=g)
    out = y0.clone().to(DEV)
    bv.conv1d_h3(x.to(DEV), w3, bias.to(DEV), Cout, k, d, res=res.to(DEV), out=out, acc_mode=2, div=3.0)
    ref2 = (y0.double() + (ref + res.double())) / 3.0
    assert float((out.cpu().double() - ref2).abs().max()) < 5e-6 * max(1.0, float(ref2.abs().max()))


def test_conv1d_h3_ragged_rows_and_tiny_values():
    from indextts_amd import bigvgan as bv
    g = torch.Generator().manual_seed(5)
    B, C, T, k, d = 3, 64, 400, 7, 5
    x = torch.randn(B, C, T, generator=g)
    x[0, :, 100:200] *= 1e-6                       # below the f16 normal range: the low part carries them (or they are negligible)
    w = torch.randn(C, C, k, generator=g) / (C * k) ** 0.5
    bias = torch.randn(C, generator=g) * 0.1
    lens = [400, 131, 17]
    w3 = bv.pack_conv1d_h3_weight(w).to(DEV)
    y = torch.full((B, C, T), 7.0, device=DEV)
    bv.conv1d_h3(x.to(DEV), w3, bias.to(DEV), C, k, d, lens=lens, out=y)
    y = y.cpu()
    for b, n in enumerate(lens):
        ref = F.conv1d(x[b:b + 1, :, :n].double(), w.double(), bias.double(), dilation=d, padding=(k - 1) // 2 * d)
        assert float((y[b:b + 1, :, :n].double() - ref).abs().max()) < 5e-6 * float(ref.abs().max())
        assert bool((y[b, :, n:] == 7.0).all())    # frames beyond a row's length are not written


@pytest.mark.parametrize("tag", ["small", "loud", "mid", "full"])
def test_generator_f16x3_vs_reference_golden(golden_dir, tag):
    from indextts_amd import bigvgan as bv
    z = np.load(os.path.join(golden_dir, f"bigvgan_gen_{tag}.npz"))
    h = dict(O.V2_HPARAMS, upsample_initial_channel=int(z["upsample_initial_channel"]))
    sd = O.synth_weights(h, seed=int(z["seed"]), post_gain=float(z["post_gain"]))
    errs = {}
    for mode in ("f32", "f16x3"):
        m = bv.BigVGAN(h, conv_mode=mode, h3_min_channels=32)      # every resblock whose width is a multiple of 32
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        wav = m(torch.from_numpy(z["mel"]).to(DEV)).cpu().numpy()
        assert wav.shape == z["wav"].shape
        errs[mode] = rms(wav - z["wav"])
    print(f"{tag}: rms err vs the reference class  f32 {errs['f32']:.3e}  f16x3 {errs['f16x3']:.3e}  (signal rms {rms(z['wav']):.3f})")
    assert errs["f16x3"] <= 1e-5


def test_generator_f16x3_ragged_rows_equal_solo():
    from indextts_amd import bigvgan as bv
    h = dict(O.V2_HPARAMS, upsample_initial_channel=512)
    sd = O.synth_weights(h, seed=11)
    m = bv.BigVGAN(h, conv_mode="f16x3", h3_min_channels=32)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    g = torch.Generator().manual_seed(2)
    mel = torch.randn(3, 80, 40, generator=g)
    lens = [40, 23, 7]
    for b, n in enumerate(lens):
        mel[b, :, n:] = 0
    wav = m(mel.to(DEV), lens=torch.tensor(lens)).cpu()
    for b, n in enumerate(lens):
        solo = m(mel[b:b + 1, :, :n].to(DEV)).cpu()
        assert float((wav[b, ..., : n * 256] - solo[0]).abs().max()) <= 1e-6


def test_generator_f16x3_out_of_range_activation_raises():
    from indextts_amd import _lib, bigvgan as bv
    h = dict(O.V2_HPARAMS, upsample_initial_channel=512)
    sd = O.synth_weights(h, seed=11)
    m = bv.BigVGAN(h, conv_mode="f16x3", h3_min_channels=32)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    mel = torch.randn(1, 80, 12, generator=torch.Generator().manual_seed(3))
    m(mel.to(DEV))                                               # in range: fine
    with pytest.raises(_lib.HipEngineError):
        m((mel * 1e7).to(DEV))                                   # drives the resblock inputs beyond 65504
    m(mel.to(DEV))                                               # the flag was cleared: the model keeps working
