"""Mint tests/golden/ecapa.npz: the reference's OWN `ECAPA_TDNN` (indextts/BigVGAN/ECAPA_TDNN.py, the speaker encoder inside the IndexTTS-1 /
1.5 vocoder, models.py:191) in eval mode, loaded strictly with oracle/ecapa_oracle.py's seeded weights (non-trivial BatchNorm statistics) and
run on seeded mel matrices of three lengths.  Two sizes: the shipped architecture (100 mels -> 512, C = 512) and a narrow one (C = 64) whose
outputs travel as the engine fixture.  The oracle and the engine are tested against these outputs."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import ecapa_oracle as EO  # noqa: E402

GOLD = os.path.join(HERE, "..", "tests", "golden")
SMALL = EO.EcapaCfg(input_size=100, lin_neurons=64, channels=64, attention_channels=32, se_channels=32, res2net_scale=8)
FULL = EO.EcapaCfg()
LENGTHS = (23, 150)


def main():
    sys.path.insert(0, HERE)
    import ref_shim_s2mel as R           # torchaudio / librosa stubs; the reference is imported only when minting
    R.install()
    from indextts.BigVGAN.ECAPA_TDNN import ECAPA_TDNN
    out = {}
    g = torch.Generator().manual_seed(43)
    for tag, cfg in (("small", SMALL), ("full", FULL)):
        sd = EO.synth_weights(cfg)
        m = ECAPA_TDNN(cfg.input_size, lin_neurons=cfg.lin_neurons, channels=[cfg.channels] * 4 + [3 * cfg.channels],
                       attention_channels=cfg.attention_channels, se_channels=cfg.se_channels, res2net_scale=cfg.res2net_scale).eval()
        m.load_state_dict(sd, strict=True)
        for i, T in enumerate(LENGTHS):
            mel = torch.randn(1, T, cfg.input_size, generator=g) * 2.0 - 4.0
            with torch.no_grad():
                ref = m(mel)
                mine = EO.ecapa(sd, cfg, mel)
            print(f"{tag} T={T}: embedding {tuple(ref.shape)} rms {float(ref.pow(2).mean().sqrt()):.3f}, oracle vs reference max|d| {float((mine - ref).abs().max()):.2e}")
            out[f"{tag}_mel{i}"], out[f"{tag}_emb{i}"] = mel[0].numpy(), ref[0, 0].numpy()
    with torch.no_grad():                # a batch of two equals the rows alone (lengths=None: no cross-row statistics in eval mode)
        sd = EO.synth_weights(SMALL)
        two = torch.randn(2, 40, 100, generator=g)
        print("batch rows vs alone:", float((EO.ecapa(sd, SMALL, two)[1] - EO.ecapa(sd, SMALL, two[1:2])[0]).abs().max()))
    np.savez_compressed(os.path.join(GOLD, "ecapa.npz"), **out)
    print("wrote ecapa.npz", os.path.getsize(os.path.join(GOLD, "ecapa.npz")))


if __name__ == "__main__":
    main()
