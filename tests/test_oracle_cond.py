"""oracle/cond_oracle.py (Conformer encoder + Perceiver resampler conditioning) against tests/golden/cond.npz:
outputs of the reference's own ConformerEncoder / PerceiverResampler classes on the oracle's seeded weights
(tools/make_golden_cond.py).  Bars: 1e-5 absolute (measured 0 / 8e-7)."""
import os

import numpy as np
import torch

from oracle import cond_oracle as CO
from tools.make_golden_cond import CCFG, ECFG, EPCFG, PCFG, weights


def test_conditioning_oracle_matches_reference_classes(golden_dir):
    z = np.load(os.path.join(golden_dir, "cond.npz"))
    sd = weights()
    feats, lens = torch.from_numpy(z["feats"]), torch.from_numpy(z["lens"])
    with torch.no_grad():
        h, mask = CO.conformer_encoder(sd, CCFG, feats, lens, "conditioning_encoder.")
        conds = CO.conditioning(sd, CCFG, PCFG, feats, lens, "conditioning_encoder.", "perceiver_encoder.")
        ev = CO.get_emovec(sd, ECFG, EPCFG, torch.from_numpy(z["emo_feats"]), torch.from_numpy(z["emo_lens"]))
        merged = CO.merge_emovec(sd, ECFG, EPCFG, feats[:2, :29], torch.from_numpy(z["emo_feats"]), torch.tensor([29, 23]),
                                 torch.from_numpy(z["emo_lens"]), 0.6)
    assert h.shape == z["enc_out"].shape == (3, 20, 64) and np.array_equal(mask.numpy(), z["enc_mask"])
    assert mask.sum(-1).flatten().tolist() == [20, 11, 3]                      # mask[:, :, 2::2] of lengths 41 / 23 / 8
    for got, key in ((h, "enc_out"), (conds, "conds"), (ev, "emovec"), (merged, "merged")):
        assert float(np.abs(got.numpy() - z[key]).max()) <= 1e-5, key
    assert float(np.abs(z["conds"]).mean()) > 0.1
    # every prompt alone (the pipeline's call pattern, and what the packed engine layout computes for every row of a batch)
    for b in range(3):
        n = int(lens[b])
        with torch.no_grad():
            hb, _ = CO.conformer_encoder(sd, CCFG, feats[b:b + 1, :n], lens[b:b + 1], "conditioning_encoder.")
            cb = CO.conditioning(sd, CCFG, PCFG, feats[b:b + 1, :n], lens[b:b + 1], "conditioning_encoder.", "perceiver_encoder.")
        assert float(np.abs(hb[0].numpy() - z[f"enc_out_alone{b}"]).max()) <= 1e-5
        assert float(np.abs(cb[0].numpy() - z[f"conds_alone{b}"]).max()) <= 1e-5
    # the reference's batch-composition dependence is real: a padded row differs from its alone result
    assert float(np.abs(z["enc_out"][1, :11] - z["enc_out_alone1"]).max()) > 1e-3


def test_padding_rows_do_not_leak_into_valid_rows():
    """a row's conditioning latents depend on its own valid frames only (key masks in both attention stacks, zeroed pad in the conv)"""
    sd = weights()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 31, CCFG.input_size, generator=g)
    lens = torch.tensor([31, 17])
    with torch.no_grad():
        a = CO.conditioning(sd, CCFG, PCFG, x, lens, "conditioning_encoder.", "perceiver_encoder.")
        x2 = x.clone()
        x2[1, 17:] = 9.0
        b = CO.conditioning(sd, CCFG, PCFG, x2, lens, "conditioning_encoder.", "perceiver_encoder.")
    assert float((a - b).abs().max()) <= 1e-5
