"""CPU: (1) oracle/ecapa_oracle.py against tests/golden/ecapa.npz = the reference's own ECAPA_TDNN class (tools/make_golden_ecapa.py);
(2) the HOST logic of indextts_amd/ecapa.py (weight folding, reflected row gathers, Res2Net chaining, squeeze-excite, attentive pooling) with the
C-ABI wrappers replaced by torch stand-ins -- test infrastructure only; the kernels themselves run in tests/test_gpu_ecapa.py."""
import os

import numpy as np
import pytest
import torch

from oracle import ecapa_oracle as EO
from tools.make_golden_ecapa import FULL, LENGTHS, SMALL


class TorchOps:
    """torch restatement of the unit ops ecapa.py calls (indextts_amd/{codec,cond,campplus,ecapa}.py wrappers)"""

    def linear(self, x, w, b, n_out):                        # `w` is the raw [out][k] matrix here (pack_gemm_weight patched to identity)
        y = x @ w.t()
        return y if b is None else y + b

    def act_(self, x, mode):
        return {0: torch.relu, 1: torch.nn.functional.silu, 2: torch.tanh}[mode](x)

    def affine(self, x, ld_x, C, scale, shift, relu=True):
        y = x[:, :C] * scale + shift
        return torch.relu(y) if relu else y

    def gate_(self, y, g):
        return y * torch.sigmoid(g)

    def add_(self, x, y):
        return x + y

    def attnstats(self, x, logits=None, eps=1e-12):
        a = torch.full_like(x, 1.0 / x.shape[0]) if logits is None else torch.softmax(logits, dim=0)
        mean = (a * x).sum(0, keepdim=True)
        return torch.cat([mean, torch.sqrt((a * (x - mean) ** 2).sum(0, keepdim=True).clamp(eps))], 1)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ecapa.npz"))


def test_oracle_equals_reference_class(gold):
    for tag, cfg in (("small", SMALL), ("full", FULL)):
        sd = EO.synth_weights(cfg)
        for i, T in enumerate(LENGTHS):
            e = EO.ecapa(sd, cfg, torch.from_numpy(gold[f"{tag}_mel{i}"])[None])
            assert e.shape == (1, 1, cfg.lin_neurons)
            assert float((e[0, 0] - torch.from_numpy(gold[f"{tag}_emb{i}"])).abs().max()) <= 1e-4 * float(np.abs(gold[f"{tag}_emb{i}"]).max())


def test_host_logic_on_torch_ops(gold, monkeypatch):
    from indextts_amd import cond, ecapa
    monkeypatch.setattr(cond, "pack_gemm_weight", lambda w, prec, transposed=False: w if transposed else w.t().contiguous())
    for tag, cfg in (("small", SMALL), ("full", FULL)):
        m = ecapa.ECAPA_TDNN(cfg.input_size, device="cpu", lin_neurons=cfg.lin_neurons, channels=[cfg.channels] * 4 + [3 * cfg.channels],
                             attention_channels=cfg.attention_channels, se_channels=cfg.se_channels, res2net_scale=cfg.res2net_scale, ops=TorchOps())
        m.load_state_dict({"speaker_encoder." + k: v for k, v in EO.synth_weights(cfg).items()}, prefix="speaker_encoder.")
        for i, T in enumerate(LENGTHS):
            e = m(torch.from_numpy(gold[f"{tag}_mel{i}"])[None])
            ref = torch.from_numpy(gold[f"{tag}_emb{i}"])
            err = float((e[0, 0] - ref).abs().max())
            print(f"{tag} T={T}: host logic on torch ops vs the reference class {err:.2e} (scale {float(ref.abs().max()):.1f})")
            assert e.shape == (1, 1, cfg.lin_neurons) and err <= 1e-4 * float(ref.abs().max())
    two = torch.from_numpy(np.stack([gold["full_mel0"], gold["full_mel0"][::-1].copy()]))
    assert float((m(two)[0] - m(two[:1])[0]).abs().max()) == 0.0
    with pytest.raises(NotImplementedError):
        m(two, lengths=torch.ones(2))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 100))                                # three frames: the k = 5 reflect padding needs more
    with pytest.raises(NotImplementedError):
        ecapa.ECAPA_TDNN(100, device="cpu", kernel_sizes=(5, 3, 3, 3, 3), ops=TorchOps())
