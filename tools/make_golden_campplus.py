"""Mint tests/golden/campplus.npz: the reference's OWN `CAMPPlus(feat_dim=80, embedding_size=192)` class
(indextts/s2mel/modules/campplus/DTDNN.py, imported from /root/reference, eval mode as in infer_v2_5.py:218-221) loaded strictly
with oracle/campplus_oracle.py's seeded weights (non-trivial BatchNorm running statistics) and run on seeded feature matrices of
three lengths (one longer than the 100-frame pooling segment after the stride-2 TDNN).  The oracle and the engine are tested
against these outputs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import campplus_oracle as CO  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
LENGTHS = (57, 130, 263)


def main():
    sys.path.insert(0, "/root/reference")
    from indextts.s2mel.modules.campplus.DTDNN import CAMPPlus
    sd = CO.synth_weights()
    m = CAMPPlus(feat_dim=80, embedding_size=192)
    m.load_state_dict(sd, strict=True)
    m.eval()
    g = torch.Generator().manual_seed(23)
    out = {}
    for i, T in enumerate(LENGTHS):
        feats = torch.randn(1, T, 80, generator=g) * 1.5
        feats = feats - feats.mean(dim=1, keepdim=True)               # infer_v2_5.py:648
        with torch.no_grad():
            ref = m(feats)
            mine = CO.campplus(sd, feats)
        out[f"feats{i}"], out[f"style{i}"] = feats[0].numpy(), ref[0].numpy()
        print(f"T={T}: style rms {float(ref.pow(2).mean().sqrt()):.3f}, oracle vs reference max|d| {float((mine - ref).abs().max()):.2e}")
    np.savez_compressed(os.path.join(GOLD, "campplus.npz"), **out)
    print("wrote campplus.npz")


if __name__ == "__main__":
    main()
