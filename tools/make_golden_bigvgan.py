#!/usr/bin/env python3
"""Mint BigVGAN golden vectors by running the REFERENCE classes themselves.

Runs only in the build container (needs /root/reference).  Imports
`indextts.s2mel.modules.bigvgan.bigvgan.BigVGAN` (librosa stubbed: it is pulled in
only by meldataset.py via utils.py), loads the seeded synthetic weights of
`oracle.bigvgan_oracle.synth_weights`, and stores input mel + reference waveform.
Weights are NOT stored (regenerated from the seed at test time).

Also mints Activation1d goldens from the reference torch Activation1d
(alias_free_activation/torch/act.py) -- the de-facto oracle of the fused kernel.

Usage: python tools/make_golden_bigvgan.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

for name in ("librosa", "librosa.util", "librosa.filters"):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.normalize = lambda *a, **k: None
        m.mel = lambda *a, **k: None
        sys.modules[name] = m
sys.modules["librosa"].util = sys.modules["librosa.util"]
sys.modules["librosa"].filters = sys.modules["librosa.filters"]

from indextts.s2mel.modules.bigvgan import bigvgan as ref_bigvgan  # noqa: E402
from indextts.s2mel.modules.bigvgan.alias_free_activation.torch.act import Activation1d  # noqa: E402
from indextts.s2mel.modules.bigvgan import activations as ref_act  # noqa: E402
from indextts.s2mel.modules.bigvgan.env import AttrDict  # noqa: E402

from oracle import bigvgan_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def ref_model(h, sd):
    model = ref_bigvgan.BigVGAN(AttrDict(dict(h, resblock="1")))
    model.remove_weight_norm()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not [m for m in missing if "filter" not in m], missing
    return model.eval()


def main():
    torch.manual_seed(0)
    # ---- Activation1d goldens (several T incl. tiny / odd lengths) ----
    acts = {}
    for tag, (B, C, T) in {"a": (2, 5, 1), "b": (1, 3, 2), "c": (2, 4, 7), "d": (1, 6, 64), "e": (2, 3, 301)}.items():
        g = torch.Generator().manual_seed(100 + T)
        x = torch.randn(B, C, T, generator=g) * 1.5
        al = torch.rand(C, generator=g) - 0.5
        be = torch.rand(C, generator=g) - 0.5
        sb = ref_act.SnakeBeta(C, alpha_logscale=True)
        sb.alpha.data.copy_(al)
        sb.beta.data.copy_(be)
        with torch.no_grad():
            y = Activation1d(activation=sb)(x)
        acts[f"{tag}_x"], acts[f"{tag}_alpha"], acts[f"{tag}_beta"], acts[f"{tag}_y"] = (
            x.numpy(), al.numpy(), be.numpy(), y.numpy())
        yo = O.activation1d(x, al, be)
        print(f"act {tag} T={T}: oracle vs reference max|d| = {(yo - y).abs().max().item():.3e}")
    acts["filter"] = Activation1d(activation=ref_act.SnakeBeta(1)).upsample.filter.reshape(-1).numpy()
    np.savez_compressed(os.path.join(GOLD, "bigvgan_act1d.npz"), **acts)

    # ---- generator goldens ----
    cases = {
        # tag: (hparam overrides, seed, B, T_mel)
        "small": (dict(upsample_initial_channel=512), 11, 2, 9, 0.04),
        "loud": (dict(upsample_initial_channel=512), 13, 1, 5, 0.35),
        "mid": (dict(upsample_initial_channel=1024), 12, 1, 20, 0.04),
        "full": (dict(), 1234, 1, 12, 0.04),
    }
    for tag, (ov, seed, B, T, pg) in cases.items():
        h = dict(O.V2_HPARAMS, **ov)
        sd = O.synth_weights(h, seed=seed, post_gain=pg)
        model = ref_model(h, sd)
        g = torch.Generator().manual_seed(seed + 1)
        mel = torch.randn(B, h["num_mels"], T, generator=g) * 2.0 - 4.0
        with torch.no_grad():
            wav = model(mel)
            wav_o = O.bigvgan_forward(sd, mel, h)
        d = (wav_o - wav)
        print(f"gen {tag}: wav rms={wav.pow(2).mean().sqrt().item():.4f} "
              f"clamped={(wav.abs() >= 1).float().mean().item():.4f} "
              f"oracle-vs-ref rms={d.pow(2).mean().sqrt().item():.3e} max={d.abs().max().item():.3e}")
        np.savez_compressed(os.path.join(GOLD, f"bigvgan_gen_{tag}.npz"),
                            mel=mel.numpy(), wav=wav.numpy().astype(np.float32),
                            seed=np.int64(seed), post_gain=np.float64(pg), upsample_initial_channel=np.int64(h["upsample_initial_channel"]))


def make_v1():
    """IndexTTS-1 / 1.5 vocoder: the reference `indextts/BigVGAN/models.py::BigVGAN` (GPT latent in, speaker-conditioned, tanh
    out) run here with torchaudio stubbed (pulled in only by the ECAPA-TDNN speaker encoder's feature front end).  The speaker
    encoder is outside the hot path (it stays a PyTorch module in the product): the reference generator is run with its
    `speaker_encoder` replaced by a module that returns the stored embedding, so everything from `conv_pre` on is the
    reference's own code.  Upsampler kernels are the v1.5 config's [8, 8, 4, 4, 4, 4] over rates [4, 4, 4, 4, 2, 2] (two k == u
    stages)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_shim_s2mel as R
    R.install()
    from indextts.BigVGAN import models as v1

    h = dict(O.V2_HPARAMS, upsample_initial_channel=512, use_tanh_at_final=True, use_bias_at_final=True,
             upsample_rates=[4, 4, 4, 4, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4])
    seed, cond_dim, gpt_dim, B, T = 17, 64, 48, 2, 7
    sd = O.synth_weights(h, seed=seed, cond_dim=cond_dim, in_dim=gpt_dim, post_gain=0.2)
    hh = AttrDict(dict(h, resblock="1", gpt_dim=gpt_dim, feat_upsample=False, cond_d_vector_in_each_upsampling_layer=True,
                       speaker_embedding_dim=cond_dim, num_mels=100))
    model = v1.BigVGAN(hh)
    model.remove_weight_norm()
    ref_sd = model.state_dict()
    load = {k: v for k, v in sd.items() if k in ref_sd}
    missing, unexpected = model.load_state_dict(load, strict=False)
    assert not unexpected
    assert all(m.startswith("speaker_encoder.") or "filter" in m for m in missing), [m for m in missing if not m.startswith("speaker_encoder.")][:5]
    assert set(sd) - set(ref_sd) == set(), set(sd) - set(ref_sd)
    g = torch.Generator().manual_seed(seed + 1)
    latent = torch.randn(B, T, gpt_dim, generator=g)
    spk = torch.randn(B, cond_dim, generator=g)

    class FixedSpeaker(torch.nn.Module):
        def forward(self, mel_ref, lens=None):
            return spk.unsqueeze(1)                                # (B, 1, dim) like ECAPA_TDNN

    model.speaker_encoder = FixedSpeaker()
    model.eval()
    with torch.no_grad():
        wav, _ = model(latent, torch.zeros(B, 5, 100))
        wav_o = O.bigvgan_forward(sd, latent.transpose(1, 2), h, spk=spk.unsqueeze(-1))
    d = wav_o - wav
    print(f"gen v1: wav {tuple(wav.shape)} rms={wav.pow(2).mean().sqrt().item():.4f} oracle-vs-ref rms={d.pow(2).mean().sqrt().item():.3e} "
          f"max={d.abs().max().item():.3e}")
    np.savez_compressed(os.path.join(GOLD, "bigvgan_v1.npz"), latent=latent.numpy(), spk=spk.numpy(), wav=wav.numpy().astype(np.float32),
                        seed=np.int64(seed), post_gain=np.float64(0.2), cond_dim=np.int64(cond_dim), gpt_dim=np.int64(gpt_dim),
                        upsample_initial_channel=np.int64(512), upsample_rates=np.array(h["upsample_rates"]),
                        upsample_kernel_sizes=np.array(h["upsample_kernel_sizes"]))


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] != "v1":
        main()
    make_v1()
