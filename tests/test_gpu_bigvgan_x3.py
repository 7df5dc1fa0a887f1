"""GPU tests of the bf16 x 3 conv mode of the vocoder (`BigVGAN(conv_mode="bf16x3")`, `itts_conv1d_x3_forward`): every f32 operand carried exactly
as three bf16 planes, six exact bf16 MFMA plane products per f32 product, f32 accumulation -- the arithmetic of the flow-matching stage's fp32x3
GEMMs (tests/test_gpu_gemm_x3.py).  The bar is the one that mode was admitted under: the error against an f64 convolution is NOT ABOVE the
f32-MFMA kernel's on the same operands; the generator is held to the waveforms the REFERENCE BigVGAN class produced at the f32 path's own error."""
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


X3_CASES = [
    (96, 96, 3, 1, 130, 2),        # one co tile, one frame tile
    (192, 192, 11, 5, 260, 1),     # taps reach 25 frames out (the widest window of the generator), two frame tiles
    (96, 96, 7, 5, 515, 2),        # three frame tiles, ragged last one
    (32, 48, 3, 1, 77, 3),         # fewer n-tiles than a block stages (clamped), one K tile per tap
    (384, 384, 7, 3, 256, 1),      # exactly one frame tile, 4 co tiles
    (128, 128, 3, 3, 1000, 1),     # a width that is not a multiple of the 96-channel tile
    (1536, 1536, 3, 1, 300, 1),    # the widest stage of the production generator (48 K tiles per tap)
]


@pytest.mark.parametrize("Cin,Cout,k,d,T,B", X3_CASES)
def test_conv1d_x3_vs_f64_not_worse_than_the_f32_kernel(Cin, Cout, k, d, T, B):
    from indextts_amd import bigvgan as bv
    g = torch.Generator().manual_seed(Cin * 1000 + Cout + k)
    x = torch.randn(B, Cin, T, generator=g) * 3
    w = torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv1d(x.double(), w.double(), bias.double(), dilation=d, padding=(k - 1) // 2 * d)
    w3 = bv.pack_conv1d_x3_weight(w).to(DEV)
    y = bv.conv1d_x3(x.to(DEV), w3, bias.to(DEV), Cout, k, d).cpu()
    y32 = bv.conv1d(x.to(DEV), bv.pack_conv1d_weight(w).to(DEV), bias.to(DEV), Cout, k, d).cpu()
    scale = max(1.0, float(ref.abs().max()))
    e_x3, e_32 = float((y.double() - ref).abs().max()), float((y32.double() - ref).abs().max())
    r_x3, r_32 = rms(y.double() - ref), rms(y32.double() - ref)
    print(f"x3 conv {Cin}x{Cout} k{k} d{d}: vs f64 max {e_x3:.2e} rms {r_x3:.2e}   f32-MFMA kernel: max {e_32:.2e} rms {r_32:.2e}   (scale {scale:.1f})")
    assert r_x3 <= 1.05 * r_32 + 1e-9 * scale and e_x3 <= 1.25 * e_32 + 1e-7 * scale
    res = torch.randn(B, Cout, T, generator=g)
    y0 = torch.randn(B, Cout, T, generator=g)
    out = y0.clone().to(DEV)
    bv.conv1d_x3(x.to(DEV), w3, bias.to(DEV), Cout, k, d, res=res.to(DEV), out=out, acc_mode=2, div=3.0)
    ref2 = (y0.double() + (ref + res.double())) / 3.0
    assert float((out.cpu().double() - ref2).abs().max()) < 5e-6 * max(1.0, float(ref2.abs().max()))


def test_conv1d_x3_ragged_rows_tiny_and_huge_values():
    """rows of different lengths (frames beyond a row's length are neither read nor written), values far below and above the f16 range (the planes
    are bf16: the f32 exponent range), and bit equality of a row with the same row alone"""
    from indextts_amd import bigvgan as bv
    g = torch.Generator().manual_seed(5)
    B, C, T, k, d = 3, 96, 400, 7, 5
    x = torch.randn(B, C, T, generator=g)
    x[0, :, 100:200] *= 1e-9
    x[0, :, 250:300] *= 1e9
    w = torch.randn(C, C, k, generator=g) / (C * k) ** 0.5
    bias = torch.randn(C, generator=g) * 0.1
    lens = [400, 131, 17]
    w3 = bv.pack_conv1d_x3_weight(w).to(DEV)
    y = torch.full((B, C, T), 7.0, device=DEV)
    bv.conv1d_x3(x.to(DEV), w3, bias.to(DEV), C, k, d, lens=lens, out=y)
    y = y.cpu()
    for b, n in enumerate(lens):
        ref = F.conv1d(x[b:b + 1, :, :n].double(), w.double(), bias.double(), dilation=d, padding=(k - 1) // 2 * d)
        assert float((y[b:b + 1, :, :n].double() - ref).abs().max()) < 2e-6 * float(ref.abs().max())
        assert bool((y[b, :, n:] == 7.0).all())    # frames beyond a row's length are not written
        solo = bv.conv1d_x3(x[b:b + 1, :, :n].contiguous().to(DEV), w3, bias.to(DEV), C, k, d).cpu()
        assert torch.equal(solo[0], y[b, :, :n])


@pytest.mark.parametrize("tag", ["small", "loud", "mid", "full"])
def test_generator_bf16x3_vs_reference_golden(golden_dir, tag):
    from indextts_amd import bigvgan as bv
    z = np.load(os.path.join(golden_dir, f"bigvgan_gen_{tag}.npz"))
    h = dict(O.V2_HPARAMS, upsample_initial_channel=int(z["upsample_initial_channel"]))
    sd = O.synth_weights(h, seed=int(z["seed"]), post_gain=float(z["post_gain"]))
    errs = {}
    for mode in ("f32", "bf16x3"):
        m = bv.BigVGAN(h, conv_mode=mode, h3_min_channels=32)      # every resblock whose width is a multiple of 32
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        wav = m(torch.from_numpy(z["mel"]).to(DEV)).cpu().numpy()
        assert wav.shape == z["wav"].shape
        errs[mode] = rms(wav - z["wav"])
    print(f"{tag}: rms err vs the reference class  f32 {errs['f32']:.3e}  bf16x3 {errs['bf16x3']:.3e}  (signal rms {rms(z['wav']):.3f})")
    assert errs["bf16x3"] <= max(1e-6, 1.2 * errs["f32"])


def test_generator_bf16x3_ragged_rows_equal_solo():
    from indextts_amd import bigvgan as bv
    h = dict(O.V2_HPARAMS, upsample_initial_channel=512)
    sd = O.synth_weights(h, seed=11)
    m = bv.BigVGAN(h, conv_mode="bf16x3", h3_min_channels=32)
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


@pytest.mark.parametrize("frames,lens", [(40, [40, 23, 7]), (67, [67, 1, 66]), (5, [5, 3, 2])])
def test_activation_written_as_operand_planes_is_bit_identical(frames, lens):
    """`voc_act_planes` (default): the activation in front of an x3 conv writes the conv's three bf16 operand planes itself (aa_act_planes_kernel)
    instead of an f32 tensor + the split pass.  Same arithmetic per element -> the waveform is bit-identical to the two-kernel path, ragged rows,
    frame counts that are not a multiple of the 128-frame tile included."""
    from indextts_amd import _lib, bigvgan as bv
    h = dict(O.V2_HPARAMS, upsample_initial_channel=1536)        # stage widths 768 ... 24: 768 / 384 / 192 / 96 / (32-multiples only) on the x3 kernel
    sd = O.synth_weights(h, seed=5)
    m = bv.BigVGAN(h, conv_mode="bf16x3", h3_min_channels=32)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    g = torch.Generator().manual_seed(frames)
    mel = torch.randn(3, 80, frames, generator=g)
    for b, n in enumerate(lens):
        mel[b, :, n:] = 0
    out = {}
    try:
        for v in (1, 0):
            _lib.set_option("voc_act_planes", v)
            out[v] = m(mel.to(DEV), lens=torch.tensor(lens)).cpu()
    finally:
        _lib.reset_options()
    for b, n in enumerate(lens):
        a, c = out[1][b, ..., : n * 256], out[0][b, ..., : n * 256]
        assert torch.isfinite(a).all() and float(a.abs().max()) > 0
        assert torch.equal(a, c), f"row {b}: max |d| {float((a - c).abs().max()):.3e}"
