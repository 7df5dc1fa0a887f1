"""Mint tests/golden/w2vbert.npz: the installed `transformers.Wav2Vec2BertModel` (the third-party class behind the reference's
`semantic_model`, indextts/infer_v2_5.py:171-176,282-290) built with a small config of the w2v-bert-2.0 architecture (relative_key
positions, causal depthwise conv, swish), loaded with oracle/w2vbert_oracle.py's seeded weights and run with
`output_hidden_states=True` on a padded batch of two feature matrices.  The oracle and the engine are tested against it."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import w2vbert_oracle as WO  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
CFG = WO.W2VBertCfg(hidden_size=64, num_hidden_layers=4, num_attention_heads=2, intermediate_size=128, feature_projection_input_dim=32,
                    left_max_position_embeddings=6, right_max_position_embeddings=2, conv_depthwise_kernel_size=7)
LAYER = 3                                  # the reference taps hidden_states[17] of 24; here [3] of 4


def main():
    from transformers import Wav2Vec2BertConfig, Wav2Vec2BertModel
    hc = Wav2Vec2BertConfig(hidden_size=CFG.hidden_size, num_hidden_layers=CFG.num_hidden_layers, num_attention_heads=CFG.num_attention_heads,
                            intermediate_size=CFG.intermediate_size, feature_projection_input_dim=CFG.feature_projection_input_dim,
                            position_embeddings_type="relative_key", left_max_position_embeddings=CFG.left_max_position_embeddings,
                            right_max_position_embeddings=CFG.right_max_position_embeddings,
                            conv_depthwise_kernel_size=CFG.conv_depthwise_kernel_size, hidden_act="swish", add_adapter=False,
                            apply_spec_augment=False, layerdrop=0.0, layer_norm_eps=CFG.layer_norm_eps)
    m = Wav2Vec2BertModel(hc).eval()
    sd = WO.synth_weights(CFG)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("masked_spec_embed" in k for k in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(31)
    feats = torch.randn(2, 37, CFG.feature_projection_input_dim, generator=g)
    mask = torch.ones(2, 37, dtype=torch.long)
    mask[1, 22:] = 0
    mean, std = 0.3 * torch.randn(CFG.hidden_size, generator=g), 0.5 + torch.rand(CFG.hidden_size, generator=g)
    with torch.no_grad():
        hs = m(input_features=feats, attention_mask=mask, output_hidden_states=True).hidden_states
        emb = (hs[LAYER] - mean) / std
        alone = m(input_features=feats[1:2, :22], attention_mask=mask[1:2, :22], output_hidden_states=True).hidden_states[LAYER]
        mine = WO.hidden_states(sd, CFG, feats, mask)
    print("oracle vs transformers, per hidden state:", [f"{float((a - b)[mask.bool()].abs().max()):.1e}" for a, b in zip(mine, hs)])
    print("row 1 inside the padded batch vs alone:", float((hs[LAYER][1, :22] - alone[0]).abs().max()))
    np.savez_compressed(os.path.join(GOLD, "w2vbert.npz"), feats=feats.numpy(), mask=mask.numpy(), mean=mean.numpy(), std=std.numpy(),
                        emb=emb.numpy(), last=hs[-1].numpy(), h1=hs[1].numpy())
    print("wrote w2vbert.npz")


if __name__ == "__main__":
    main()
