"""GPU parity of the ECAPA-TDNN speaker encoder of the v1 / v1.5 vocoder (indextts_amd/ecapa.py) on the HIP engine,
through the C ABI, against tests/golden/ecapa.npz = outputs of the REFERENCE's own ECAPA_TDNN class on the oracle's seeded weights
(tools/make_golden_ecapa.py), at a narrow width and at the shipped one (100 mels -> 512, C = 512).  Exact-f32 unit ops through ~40 layers:
bar 2e-4 of the embedding's largest component (the host logic on torch ops lands at 1e-6 relative, tests/test_host_ecapa.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import bigvgan_oracle as BO
from oracle import ecapa_oracle as EO
from tools.make_golden_ecapa import FULL, LENGTHS, SMALL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_attnstats_and_tanh_vs_torch():
    from indextts_amd.ecapa import _EOps
    ops = _EOps(DEV)
    g = torch.Generator().manual_seed(6)
    x, lg = torch.randn(301, 192, generator=g) * 2 + 1, torch.randn(301, 192, generator=g) * 3
    for logits in (None, lg):
        st = ops.attnstats(x.to(DEV), None if logits is None else logits.to(DEV)).cpu()
        a = torch.full_like(x, 1.0 / 301) if logits is None else torch.softmax(logits, dim=0)
        mean = (a * x).sum(0, keepdim=True)
        ref = torch.cat([mean, torch.sqrt((a * (x - mean) ** 2).sum(0, keepdim=True).clamp(1e-12))], 1)
        assert st.shape == (1, 384) and float((st - ref).abs().max()) <= 2e-5
    one = ops.attnstats(x[:1].to(DEV)).cpu()                          # a single frame: the clamp keeps the root finite
    assert float((one[0, :192] - x[0]).abs().max()) <= 1e-6 and float(one[0, 192:].max()) <= 1e-5
    t = ops.act_(x.clone().to(DEV), 2).cpu()
    assert float((t - torch.tanh(x)).abs().max()) <= 1e-6


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ecapa.npz"))


@pytest.mark.parametrize("tag", ["small", "full"])
def test_ecapa_vs_reference_class(gold, tag):
    from indextts_amd.ecapa import ECAPA_TDNN
    cfg = SMALL if tag == "small" else FULL
    m = ECAPA_TDNN(cfg.input_size, device=DEV, lin_neurons=cfg.lin_neurons, channels=[cfg.channels] * 4 + [3 * cfg.channels],
                   attention_channels=cfg.attention_channels, se_channels=cfg.se_channels, res2net_scale=cfg.res2net_scale)
    m.load_state_dict(EO.synth_weights(cfg)).eval()
    for i, T in enumerate(LENGTHS):
        e = m(torch.from_numpy(gold[f"{tag}_mel{i}"])[None]).cpu()
        ref = torch.from_numpy(gold[f"{tag}_emb{i}"])
        err = float((e[0, 0] - ref).abs().max())
        print(f"ECAPA-TDNN {tag} T={T}: max|d| vs the reference class {err:.2e} (largest component {float(ref.abs().max()):.1f})")
        assert e.shape == (1, 1, cfg.lin_neurons) and err <= 2e-4 * float(ref.abs().max())
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 50, cfg.input_size), lengths=torch.ones(1))


def test_v1_vocoder_builds_its_speaker_encoder_from_the_checkpoint(gold, golden_dir):
    """A v1 / v1.5 vocoder state dict with `speaker_encoder.*` tensors: BigVGAN builds the engine ECAPA-TDNN itself, `model(latent, mel_ref)` equals
    `model(latent, speaker_embedding=<the reference class's embedding>)`."""
    from indextts_amd import bigvgan as bv
    z = np.load(os.path.join(golden_dir, "bigvgan_v1.npz"))
    h = dict(BO.V2_HPARAMS, upsample_initial_channel=int(z["upsample_initial_channel"]), use_tanh_at_final=True, use_bias_at_final=True,
             upsample_rates=[int(v) for v in z["upsample_rates"]], upsample_kernel_sizes=[int(v) for v in z["upsample_kernel_sizes"]])
    cd, gd = int(z["cond_dim"]), int(z["gpt_dim"])
    assert cd == SMALL.lin_neurons
    sd = BO.synth_weights(h, seed=int(z["seed"]), cond_dim=cd, in_dim=gd, post_gain=float(z["post_gain"]))
    sd.update({"speaker_encoder." + k: v for k, v in EO.synth_weights(SMALL).items()})
    m = bv.BigVGAN(h, cond_dim=cd, in_channels=gd)
    skipped = m.load_state_dict(sd)
    assert any(k.startswith("speaker_encoder.") for k in skipped)
    m.to(DEV).eval()
    mel_ref = torch.from_numpy(gold["small_mel1"])[None].to(DEV)             # (1, T_ref, 100): `cond_mel.transpose(1, 2)` of infer.py:647
    emb = m.speaker_embedding(mel_ref).cpu()
    ref = torch.from_numpy(gold["small_emb1"])
    assert emb.shape == (1, cd) and float((emb[0] - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
    lat = torch.from_numpy(z["latent"][:1]).to(DEV)
    a, _ = m(lat, mel_ref)
    b, _ = m(lat, speaker_embedding=ref[None].to(DEV))
    assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-3                 # the embeddings differ by ~1e-6 of their scale
