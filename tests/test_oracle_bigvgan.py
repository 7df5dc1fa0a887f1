"""Pin the BigVGAN CPU oracle against fixtures minted from the reference classes (tools/make_golden_bigvgan.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import bigvgan_oracle as O


def rms(x):
    return float(np.sqrt(np.mean(np.square(np.asarray(x, dtype=np.float64)))))


def test_filter_matches_reference_buffer(golden_dir):
    z = np.load(os.path.join(golden_dir, "bigvgan_act1d.npz"))
    np.testing.assert_allclose(O.default_filter().numpy(), z["filter"], rtol=0, atol=1e-8)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e"])
def test_activation1d_vs_reference(golden_dir, tag):
    """T=1, 2, 7 (shorter than the 12-tap support: all-replicate padding), 64, 301."""
    z = np.load(os.path.join(golden_dir, "bigvgan_act1d.npz"))
    y = O.activation1d(torch.from_numpy(z[f"{tag}_x"]), torch.from_numpy(z[f"{tag}_alpha"]),
                       torch.from_numpy(z[f"{tag}_beta"]))
    np.testing.assert_allclose(y.numpy(), z[f"{tag}_y"], rtol=0, atol=5e-6)


@pytest.mark.parametrize("tag", ["small", "loud", "mid", "full"])
def test_generator_vs_reference(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, f"bigvgan_gen_{tag}.npz"))
    h = dict(O.V2_HPARAMS, upsample_initial_channel=int(z["upsample_initial_channel"]))
    sd = O.synth_weights(h, seed=int(z["seed"]), post_gain=float(z["post_gain"]))
    with torch.no_grad():
        wav = O.bigvgan_forward(sd, torch.from_numpy(z["mel"]), h).numpy()
    assert wav.shape == z["wav"].shape and wav.shape[-1] == 256 * z["mel"].shape[-1]
    assert rms(wav - z["wav"]) <= 1e-5                       # gate for the HIP path is 1e-4
    assert rms(z["wav"]) > 0.05                              # non-vacuous signal


def test_weight_norm_folding():
    g = torch.Generator().manual_seed(3)
    v = torch.randn(6, 4, 3, generator=g)
    gg = torch.rand(6, 1, 1, generator=g) + 0.5
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(4, 6, 3))
    conv.weight_v.data.copy_(v)
    conv.weight_g.data.copy_(gg)
    sd = O.fold_weight_norm({"c.weight_g": gg, "c.weight_v": v, "c.bias": torch.zeros(6)})
    assert set(sd) == {"c.weight", "c.bias"}
    x = torch.randn(1, 4, 9, generator=g)
    y_ref = conv(x)
    y = torch.nn.functional.conv1d(x, sd["c.weight"], conv.bias)
    assert torch.allclose(y, y_ref, atol=1e-6)


def test_v1_variant_runs():
    """v1 generator: latent input, speaker conditioning adds, tanh epilogue (indextts/BigVGAN/models.py:201-250)."""
    h = dict(O.V2_HPARAMS, upsample_initial_channel=512, use_tanh_at_final=True, use_bias_at_final=True,
             upsample_rates=[4, 4, 4, 4, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4])
    sd = O.synth_weights(h, seed=5, cond_dim=16, in_dim=24)
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(2, 24, 3, generator=g)
    spk = torch.randn(2, 16, 1, generator=g)
    with torch.no_grad():
        wav = O.bigvgan_forward(sd, lat, h, spk=spk)
    assert wav.shape == (2, 1, 3 * 1024) and float(wav.abs().max()) <= 1.0


def test_v1_generator_matches_reference_fixture(golden_dir):
    """tests/golden/bigvgan_v1.npz = the reference's own `indextts/BigVGAN/models.py::BigVGAN` (IndexTTS-1 / 1.5: GPT latent in,
    speaker conditioning after conv_pre and every upsampler, tanh out, upsampler kernels [8,8,4,4,4,4] over rates [4,4,4,4,2,2])
    run with its speaker encoder replaced by the stored embedding (tools/make_golden_bigvgan.py v1): pins the oracle's v1 branch."""
    z = np.load(os.path.join(golden_dir, "bigvgan_v1.npz"))
    h = dict(O.V2_HPARAMS, upsample_initial_channel=int(z["upsample_initial_channel"]), use_tanh_at_final=True, use_bias_at_final=True,
             upsample_rates=[int(v) for v in z["upsample_rates"]], upsample_kernel_sizes=[int(v) for v in z["upsample_kernel_sizes"]])
    sd = O.synth_weights(h, seed=int(z["seed"]), cond_dim=int(z["cond_dim"]), in_dim=int(z["gpt_dim"]), post_gain=float(z["post_gain"]))
    with torch.no_grad():
        wav = O.bigvgan_forward(sd, torch.from_numpy(z["latent"]).transpose(1, 2), h, spk=torch.from_numpy(z["spk"]).unsqueeze(-1))
    assert wav.shape == z["wav"].shape
    assert float((wav - torch.from_numpy(z["wav"])).double().pow(2).mean().sqrt()) <= 2e-6
