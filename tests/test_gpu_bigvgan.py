"""GPU parity tests: HIP BigVGAN path (through the C ABI) vs the CPU oracle and the reference-minted goldens.

Tolerance (north_star): waveform RMS error <= 1e-4 vs the reference CPU path; unit ops are held to 2e-5 max-abs
relative to the output scale.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import bigvgan_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rms(x):
    x = x.detach().double().cpu() if torch.is_tensor(x) else torch.as_tensor(np.asarray(x)).double()
    return float(x.pow(2).mean().sqrt())


@pytest.fixture(scope="module")
def bv():
    from indextts_amd import bigvgan
    return bigvgan


def test_library_loaded_and_sees_gpu():
    from indextts_amd import _lib
    assert _lib.lib().itts_device_count() >= 1


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e"])
def test_aa_act_vs_reference_golden(bv, golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "bigvgan_act1d.npz"))
    f = torch.from_numpy(z["filter"])
    y = bv.anti_alias_activation(torch.from_numpy(z[f"{tag}_x"]).to(DEV), f, f, torch.from_numpy(z[f"{tag}_alpha"]),
                                 torch.from_numpy(z[f"{tag}_beta"]))
    np.testing.assert_allclose(y.cpu().numpy(), z[f"{tag}_y"], rtol=0, atol=1e-5)


def test_aa_act_multi_tile_and_ragged(bv):
    g = torch.Generator().manual_seed(7)
    B, C, T = 3, 5, 2500                     # 3 tiles of 1024
    x = torch.randn(B, C, T, generator=g) * 2
    al, be = torch.rand(C, generator=g) - 0.5, torch.rand(C, generator=g) - 0.5
    f = O.default_filter()
    y = bv.anti_alias_activation(x.to(DEV), f, f, al, be).cpu()
    ref = O.activation1d(x, al, be)
    assert (y - ref).abs().max() < 2e-5
    lens = [2500, 1024, 37]
    yr = bv.anti_alias_activation(x.to(DEV), f, f, al, be, lens=lens).cpu()
    for b, n in enumerate(lens):
        ref_b = O.activation1d(x[b:b + 1, :, :n], al, be)
        assert (yr[b:b + 1, :, :n] - ref_b).abs().max() < 2e-5


CONV_CASES = [
    # Cin, Cout, k, d, T, B
    (64, 64, 3, 1, 300, 2),
    (96, 96, 7, 3, 300, 2),        # 3 co-subtiles config
    (24, 24, 11, 5, 700, 2),       # 1 co-subtile (padded), max halo, multiple time tiles
    (48, 48, 3, 5, 130, 1),        # 2 co-subtiles (padded)
    (80, 160, 7, 1, 50, 2),        # conv_pre-like: C_in not a multiple of the 32-channel chunk
    (192, 192, 11, 3, 260, 1),     # 6 co-subtiles: ragged last co tile of the 128-row config
    (40, 200, 5, 2, 129, 3),
]


@pytest.mark.parametrize("Cin,Cout,k,d,T,B", CONV_CASES)
def test_conv1d_vs_torch(bv, Cin, Cout, k, d, T, B):
    g = torch.Generator().manual_seed(Cin * 1000 + Cout + k)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv1d(x, w, bias, dilation=d, padding=(k - 1) // 2 * d)
    wp = bv.pack_conv1d_weight(w).to(DEV)
    y = bv.conv1d(x.to(DEV), wp, bias.to(DEV), Cout, k, d).cpu()
    assert (y - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))
    # residual + accumulate epilogues
    res = torch.randn(B, Cout, T, generator=g)
    y0 = torch.randn(B, Cout, T, generator=g)
    out = y0.clone().to(DEV)
    bv.conv1d(x.to(DEV), wp, bias.to(DEV), Cout, k, d, res=res.to(DEV), out=out, acc_mode=2, div=3.0)
    ref2 = (y0 + (ref + res)) / 3.0
    assert (out.cpu() - ref2).abs().max() < 3e-5 * max(1.0, float(ref2.abs().max()))


def test_conv1d_ragged_rows_equal_solo(bv):
    g = torch.Generator().manual_seed(5)
    B, C, T, k, d = 3, 64, 400, 7, 5
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(C, C, k, generator=g) / (C * k) ** 0.5
    bias = torch.randn(C, generator=g) * 0.1
    lens = [400, 131, 17]
    wp = bv.pack_conv1d_weight(w).to(DEV)
    y = bv.conv1d(x.to(DEV), wp, bias.to(DEV), C, k, d, lens=lens).cpu()
    for b, n in enumerate(lens):
        ref = F.conv1d(x[b:b + 1, :, :n], w, bias, dilation=d, padding=(k - 1) // 2 * d)
        assert (y[b:b + 1, :, :n] - ref).abs().max() < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("Cin,Cout,k,u,T,B", [(128, 64, 8, 4, 70, 2), (48, 24, 4, 2, 300, 2), (64, 32, 8, 4, 1, 1),
                                               (64, 32, 4, 4, 33, 2), (32, 16, 2, 2, 5, 1), (32, 16, 12, 4, 17, 2)])
def test_conv_transpose1d_vs_torch(bv, Cin, Cout, k, u, T, B):
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cin, Cout, k, generator=g) / (Cin * k / u) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv_transpose1d(x, w, bias, stride=u, padding=(k - u) // 2)
    wp = bv.pack_convT_weight(w, u).to(DEV)
    y = bv.conv_transpose1d(x.to(DEV), wp, bias.to(DEV), Cout, k, u).cpu()
    assert y.shape == ref.shape
    assert (y - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))


def _model(bv, h, sd, **kw):
    m = bv.BigVGAN(h, **kw)
    m.load_state_dict(sd)
    return m.to(DEV).eval()


@pytest.mark.parametrize("tag", ["small", "loud", "mid", "full"])
def test_generator_vs_reference_golden(bv, golden_dir, tag):
    """HIP waveform vs the waveform the REFERENCE BigVGAN class produced on the same weights and mel."""
    z = np.load(os.path.join(golden_dir, f"bigvgan_gen_{tag}.npz"))
    h = dict(O.V2_HPARAMS, upsample_initial_channel=int(z["upsample_initial_channel"]))
    sd = O.synth_weights(h, seed=int(z["seed"]), post_gain=float(z["post_gain"]))
    m = _model(bv, h, sd)
    wav = m(torch.from_numpy(z["mel"]).to(DEV)).cpu().numpy()
    assert wav.shape == z["wav"].shape
    err = rms(wav - z["wav"])
    print(f"{tag}: rms err {err:.3e} (signal rms {rms(z['wav']):.3f})")
    assert err <= 1e-4
    assert np.abs(wav).max() <= 1.0


def test_generator_ragged_batch_rows_equal_solo(bv):
    """Batched, zero-padded rows bounded at their own length == each row alone (the B=1 reference semantics)."""
    h = dict(O.V2_HPARAMS, upsample_initial_channel=512)
    sd = O.synth_weights(h, seed=77)
    m = _model(bv, h, sd)
    g = torch.Generator().manual_seed(3)
    lens = [23, 9, 1, 16]
    mel = torch.randn(4, 80, 23, generator=g) * 2 - 4
    for b, n in enumerate(lens):
        mel[b, :, n:] = 0
    wav = m(mel.to(DEV), lens=lens).cpu()
    for b, n in enumerate(lens):
        with torch.no_grad():
            ref = O.bigvgan_forward(sd, mel[b:b + 1, :, :n], h)
        assert rms(wav[b:b + 1, :, : n * 256] - ref) <= 1e-4
        assert float(wav[b, :, n * 256:].abs().max()) == 0.0 if n < 23 else True


def test_generator_v1_vs_reference_fixture(bv, golden_dir):
    """a-13: the HIP generator against the waveform the reference's own v1 `BigVGAN` class produced (bigvgan_v1.npz, minted by
    tools/make_golden_bigvgan.py v1 with the speaker embedding handed in): 1e-4 RMS, k == u upsamplers included."""
    z = np.load(os.path.join(golden_dir, "bigvgan_v1.npz"))
    h = dict(O.V2_HPARAMS, upsample_initial_channel=int(z["upsample_initial_channel"]), use_tanh_at_final=True, use_bias_at_final=True,
             upsample_rates=[int(v) for v in z["upsample_rates"]], upsample_kernel_sizes=[int(v) for v in z["upsample_kernel_sizes"]])
    cd, gd = int(z["cond_dim"]), int(z["gpt_dim"])
    sd = O.synth_weights(h, seed=int(z["seed"]), cond_dim=cd, in_dim=gd, post_gain=float(z["post_gain"]))
    m = _model(bv, h, sd, cond_dim=cd, in_channels=gd)
    wav, _ = m(torch.from_numpy(z["latent"]).to(DEV), speaker_embedding=torch.from_numpy(z["spk"]).to(DEV))
    err = rms(wav.cpu() - torch.from_numpy(z["wav"]))
    print(f"v1 generator vs reference class: rms {err:.2e}")
    assert wav.shape == z["wav"].shape and err <= 1e-4


def test_generator_v1_variant(bv):
    """v1: latent input (B,T,D), speaker conditioning adds after conv_pre and each upsampler, tanh epilogue."""
    h = dict(O.V2_HPARAMS, upsample_initial_channel=512, use_tanh_at_final=True, use_bias_at_final=True,
             upsample_rates=[4, 4, 4, 4, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4])
    sd = O.synth_weights(h, seed=5, cond_dim=64, in_dim=48, post_gain=0.2)
    m = _model(bv, h, sd, cond_dim=64, in_channels=48)
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(2, 6, 48, generator=g)
    spk = torch.randn(2, 64, generator=g)
    wav, _ = m(lat.to(DEV), speaker_embedding=spk.to(DEV))
    with torch.no_grad():
        ref = O.bigvgan_forward(sd, lat.transpose(1, 2), h, spk=spk.unsqueeze(-1))
    assert wav.shape == ref.shape == (2, 1, 6 * 1024)
    assert rms(wav.cpu() - ref) <= 1e-4


def test_full_size_batching_invariance(bv):
    """At the BASELINE mel length (config 2: T = int(2*350*1.72) = 1204) the batched call must reproduce the
    per-row call bit-for-bit (same kernels, same tiles -> deterministic), and the waveform is bounded."""
    h = dict(O.V2_HPARAMS)
    sd = O.synth_weights(h, seed=1234)
    m = _model(bv, h, sd)
    g = torch.Generator().manual_seed(9)
    mel = torch.randn(2, 80, 1204, generator=g) * 2 - 4
    both = m(mel.to(DEV))
    solo = m(mel[1:2].to(DEV))
    assert torch.equal(both[1:2], solo)
    assert float(both.abs().max()) <= 1.0 and rms(both) > 0.05


def test_streaming_equals_one_shot(bv):
    """configs[4] (long-form, chunked vocoding): overlap-save chunks with a receptive-field halo reproduce the one-shot
    waveform (gate 1e-4 RMS; observed ~1e-7), including chunk sizes that do not divide T and a too-small halo failing."""
    h = dict(O.V2_HPARAMS, upsample_initial_channel=512)
    sd = O.synth_weights(h, seed=21)
    m = _model(bv, h, sd)
    g = torch.Generator().manual_seed(4)
    mel = (torch.randn(1, 80, 333, generator=g) * 2 - 4).to(DEV)
    full = m(mel)
    rf = m.receptive_field_frames()
    assert 20 <= rf <= 64
    for chunk in (64, 100, 333):
        st = m.forward_chunked(mel, chunk_frames=chunk)
        assert st.shape == full.shape
        assert rms(st - full) <= 1e-6, (chunk, rms(st - full))
    bad = m.forward_chunked(mel, chunk_frames=64, halo_frames=2)
    assert rms(bad - full) > 1e-4          # the halo is what makes it exact


def test_push_stream_equals_one_shot(bv):
    """C-ABI push stream (itts_bigvgan_stream_*): ragged chunk sizes, output trailing the input by the halo, flush on the
    last push -- the concatenation equals the one-shot forward (same 1e-4 RMS gate; measured bitwise-close)."""
    h = dict(O.V2_HPARAMS, upsample_initial_channel=512)
    sd = O.synth_weights(h, seed=9)
    m = _model(bv, h, sd)
    g = torch.Generator().manual_seed(4)
    T = 333
    mel = (torch.randn(1, 80, T, generator=g) * 2 - 4).to(DEV)
    ref = m(mel)
    st = m.open_stream(chunk_frames=64)
    outs, t = [], 0
    for n in (64, 10, 64, 1, 50, 64, 64, 16):
        outs.append(st.push(mel[0, :, t:t + n], last=False))
        t += n
    assert t == T
    assert sum(o.shape[-1] for o in outs) == (T - st.halo) * 256          # everything but the halo is already out
    outs.append(st.push(mel[0, :, T:T], last=True))
    got = torch.cat(outs, dim=-1)
    st.close()
    assert got.shape == ref.shape
    assert rms(got - ref) <= 1e-4
    # a stream shorter than the halo emits nothing until the flush
    st = m.open_stream(chunk_frames=32)
    a = st.push(mel[0, :, :20])
    b = st.push(mel[0, :, 20:20], last=True)
    assert a.shape[-1] == 0 and b.shape[-1] == 20 * 256
    assert rms(b - m(mel[:, :, :20].contiguous())) <= 1e-4
