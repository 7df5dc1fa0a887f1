"""GPU parity of the CAMPPlus speaker encoder (-> the `style` vector) on the HIP engine, through the C ABI,
against tests/golden/campplus.npz = outputs of the REFERENCE's own CAMPPlus class on the oracle's seeded weights
(tools/make_golden_campplus.py).  Exact-f32 unit ops through ~60 layers: bar 5e-4 absolute on outputs of RMS ~5."""
import os

import numpy as np
import pytest
import torch

from oracle import campplus_oracle as CO
from tools.make_golden_campplus import LENGTHS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 5e-4


def test_unit_ops_vs_torch():
    from indextts_amd.campplus import _COps
    ops = _COps(DEV)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(230, 48, generator=g)
    s, t = torch.randn(32, generator=g), torch.randn(32, generator=g)
    a = ops.affine(x.to(DEV), 48, 32, s.to(DEV), t.to(DEV), relu=True).cpu()
    assert float((a - torch.relu(x[:, :32] * s + t)).abs().max()) <= 1e-6
    h = x[:, :32].contiguous()
    ctx = ops.ctxpool(h.to(DEV), 100).cpu()
    ref = torch.empty_like(h)
    for s0 in range(0, 230, 100):
        ref[s0: s0 + 100] = h.mean(0) + h[s0: s0 + 100].mean(0)
    assert float((ctx - ref).abs().max()) <= 1e-5
    y, gte = torch.randn(230, 32, generator=g), torch.randn(230, 32, generator=g)
    assert float((ops.gate_(y.clone().to(DEV), gte.to(DEV)).cpu() - y * torch.sigmoid(gte)).abs().max()) <= 1e-6
    st = ops.statspool(h.to(DEV)).cpu()[0]
    assert float((st - torch.cat([h.mean(0), h.std(0, unbiased=True)])).abs().max()) <= 1e-5


def test_campplus_vs_reference_class(golden_dir):
    from indextts_amd.campplus import CAMPPlus
    z = np.load(os.path.join(golden_dir, "campplus.npz"))
    m = CAMPPlus(feat_dim=80, embedding_size=192, device=DEV)
    m.load_state_dict(CO.synth_weights())
    m.eval()
    for i, T in enumerate(LENGTHS):
        feats = torch.from_numpy(z[f"feats{i}"]).unsqueeze(0)
        y = m(feats).cpu()
        err = float((y[0] - torch.from_numpy(z[f"style{i}"])).abs().max())
        print(f"CAMPPlus T={T}: max|d| vs the reference class {err:.2e} (style rms {float(y.pow(2).mean().sqrt()):.2f})")
        assert y.shape == (1, 192) and err <= TOL
    with pytest.raises(ValueError):
        m(torch.zeros(1, 50, 64))
