"""oracle/campplus_oracle.py (CAMPPlus speaker encoder -> `style`) against tests/golden/campplus.npz:
outputs of the reference's own CAMPPlus class on the oracle's seeded weights (tools/make_golden_campplus.py).  Bar 1e-4 absolute
on outputs of RMS ~5 (measured 0)."""
import os

import numpy as np
import torch

from oracle import campplus_oracle as CO
from tools.make_golden_campplus import LENGTHS


def test_campplus_oracle_matches_reference_class(golden_dir):
    z = np.load(os.path.join(golden_dir, "campplus.npz"))
    sd = CO.synth_weights()
    assert len(sd) == 937                                           # every key of the reference module's state dict
    for i, T in enumerate(LENGTHS):
        feats = torch.from_numpy(z[f"feats{i}"]).unsqueeze(0)
        assert feats.shape == (1, T, 80)
        with torch.no_grad():
            y = CO.campplus(sd, feats)
        assert y.shape == (1, 192) and float(np.abs(y[0].numpy() - z[f"style{i}"]).max()) <= 1e-4
        assert float(np.sqrt((z[f"style{i}"] ** 2).mean())) > 1.0
