"""The w2v-bert-2.0 oracle (oracle/w2vbert_oracle.py) against tests/golden/w2vbert.npz = transformers' Wav2Vec2BertModel on the oracle's
seeded weights (tools/make_golden_w2vbert.py).  Same torch ops in the same order: bar 1e-5."""
import os

import numpy as np
import torch

from oracle import w2vbert_oracle as WO
from tools.make_golden_w2vbert import CFG, LAYER


def test_oracle_matches_transformers_class(golden_dir):
    z = np.load(os.path.join(golden_dir, "w2vbert.npz"))
    sd = WO.synth_weights(CFG)
    feats, mask = torch.from_numpy(z["feats"]), torch.from_numpy(z["mask"])
    valid = mask.bool()
    hs = WO.hidden_states(sd, CFG, feats, mask)
    assert float((hs[1] - torch.from_numpy(z["h1"]))[valid].abs().max()) <= 1e-5
    assert float((hs[-1] - torch.from_numpy(z["last"]))[valid].abs().max()) <= 1e-5
    emb = WO.get_emb(sd, CFG, feats, mask, torch.from_numpy(z["mean"]), torch.from_numpy(z["std"]), layer=LAYER)
    assert float((emb - torch.from_numpy(z["emb"]))[valid].abs().max()) <= 1e-5


def test_valid_frames_do_not_depend_on_padding(golden_dir):
    """the depthwise conv is causal and the padded keys are masked: a prompt alone == the prompt inside a padded batch"""
    z = np.load(os.path.join(golden_dir, "w2vbert.npz"))
    sd = WO.synth_weights(CFG)
    feats, mask = torch.from_numpy(z["feats"]), torch.from_numpy(z["mask"])
    n = int(mask[1].sum())
    alone = WO.hidden_states(sd, CFG, feats[1:2, :n], None)[-1]
    assert float((alone[0] - torch.from_numpy(z["last"])[1, :n]).abs().max()) <= 1e-5
