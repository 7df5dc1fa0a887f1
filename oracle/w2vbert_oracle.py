"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32, eval mode) of the w2v-bert-2.0 feature extractor behind
`IndexTTS2.get_emb` (indextts/infer_v2_5.py:282-290: `semantic_model(...).hidden_states[17]`, then (feat - mean) / std), SURVEY.md
section 8 f-3.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The model is a THIRD-PARTY dependency, absent from /root/reference: `transformers.Wav2Vec2BertModel` (the reference pins
transformers==4.52.1, pyproject.toml:61; `facebook/w2v-bert-2.0`: 24 Conformer layers, hidden 1024, 16 heads, FFN 4096, 160-dim
stacked fbank input, `relative_key` position embeddings with left / right reach 64 / 8, causal depthwise conv k = 31, swish).  Its
published algorithm (modeling_wav2vec2_bert.py: Wav2Vec2BertFeatureProjection, Wav2Vec2BertEncoderLayer, Wav2Vec2BertSelfAttention,
Wav2Vec2BertConvolutionModule, Wav2Vec2BertFeedForward, Wav2Vec2BertEncoder) is restated here; parity is pinned by
tests/golden/w2vbert.npz = outputs of the installed transformers class (5.15, same algorithm) on this module's seeded weights
(tools/make_golden_w2vbert.py), and anchored on the reference's call site above.
"""
import math
from dataclasses import dataclass
from typing import Dict, List

import torch
import torch.nn.functional as F


@dataclass
class W2VBertCfg:
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    feature_projection_input_dim: int = 160
    left_max_position_embeddings: int = 64
    right_max_position_embeddings: int = 8
    conv_depthwise_kernel_size: int = 31
    layer_norm_eps: float = 1e-5


def synth_weights(cfg: W2VBertCfg, seed: int = 29) -> Dict[str, torch.Tensor]:
    """Seeded weights under transformers' Wav2Vec2BertModel parameter names."""
    g = torch.Generator().manual_seed(seed)
    D, I, H = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads
    rn = lambda *s, fan: torch.randn(*s, generator=g) / math.sqrt(fan)
    ln = lambda p, n: {p + ".weight": 1 + 0.1 * torch.randn(n, generator=g), p + ".bias": 0.05 * torch.randn(n, generator=g)}
    sd: Dict[str, torch.Tensor] = {}
    sd.update(ln("feature_projection.layer_norm", cfg.feature_projection_input_dim))
    sd["feature_projection.projection.weight"] = rn(D, cfg.feature_projection_input_dim, fan=cfg.feature_projection_input_dim)
    sd["feature_projection.projection.bias"] = 0.05 * torch.randn(D, generator=g)
    R = cfg.left_max_position_embeddings + cfg.right_max_position_embeddings + 1
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}."
        for ff in ("ffn1", "ffn2"):
            sd.update(ln(p + ff + "_layer_norm", D))
            sd[p + ff + ".intermediate_dense.weight"], sd[p + ff + ".intermediate_dense.bias"] = rn(I, D, fan=D), 0.05 * torch.randn(I, generator=g)
            sd[p + ff + ".output_dense.weight"], sd[p + ff + ".output_dense.bias"] = rn(D, I, fan=I), 0.05 * torch.randn(D, generator=g)
        sd.update(ln(p + "self_attn_layer_norm", D))
        for n in ("q", "k", "v", "out"):
            sd[p + f"self_attn.linear_{n}.weight"], sd[p + f"self_attn.linear_{n}.bias"] = rn(D, D, fan=D), 0.05 * torch.randn(D, generator=g)
        sd[p + "self_attn.distance_embedding.weight"] = 0.5 * torch.randn(R, D // H, generator=g)
        sd.update(ln(p + "conv_module.layer_norm", D))
        sd[p + "conv_module.pointwise_conv1.weight"] = rn(2 * D, D, 1, fan=D)
        sd[p + "conv_module.depthwise_conv.weight"] = rn(D, 1, cfg.conv_depthwise_kernel_size, fan=cfg.conv_depthwise_kernel_size) * 1.5
        sd.update(ln(p + "conv_module.depthwise_layer_norm", D))
        sd[p + "conv_module.pointwise_conv2.weight"] = rn(D, D, 1, fan=D)
        sd.update(ln(p + "final_layer_norm", D))
    return sd


def _ln(sd, p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _ffn(sd, p, x):                                                  # Wav2Vec2BertFeedForward (swish)
    h = F.silu(F.linear(x, sd[p + ".intermediate_dense.weight"], sd[p + ".intermediate_dense.bias"]))
    return F.linear(h, sd[p + ".output_dense.weight"], sd[p + ".output_dense.bias"])


def _self_attn(sd, p, cfg, x, add_mask):                              # Wav2Vec2BertSelfAttention, relative_key
    B, T, D = x.shape
    H, dh = cfg.num_attention_heads, D // cfg.num_attention_heads
    lin = lambda n, t: F.linear(t, sd[p + f"linear_{n}.weight"], sd[p + f"linear_{n}.bias"])
    q = lin("q", x).view(B, T, H, dh).transpose(1, 2)
    k = lin("k", x).view(B, T, H, dh).transpose(1, 2)
    v = lin("v", x).view(B, T, H, dh).transpose(1, 2)
    scores = q @ k.transpose(-2, -1) / math.sqrt(dh)
    dist = (torch.arange(T)[None, :] - torch.arange(T)[:, None]).clamp(-cfg.left_max_position_embeddings, cfg.right_max_position_embeddings)
    pe = sd[p + "distance_embedding.weight"][dist + cfg.left_max_position_embeddings]          # (T, T, dh)
    scores = scores + torch.einsum("bhld,lrd->bhlr", q, pe) / math.sqrt(dh)
    if add_mask is not None:
        scores = scores + add_mask
    out = (torch.softmax(scores, dim=-1) @ v).transpose(1, 2).reshape(B, T, D)
    return lin("out", out)


def _conv_module(sd, p, cfg, x, mask):                                # Wav2Vec2BertConvolutionModule (causal depthwise conv)
    h = _ln(sd, p + "layer_norm", x, cfg.layer_norm_eps)
    if mask is not None:
        h = h.masked_fill(~mask.bool().unsqueeze(-1), 0.0)
    h = F.glu(F.conv1d(h.transpose(1, 2), sd[p + "pointwise_conv1.weight"]), dim=1)
    h = F.conv1d(F.pad(h, (cfg.conv_depthwise_kernel_size - 1, 0)), sd[p + "depthwise_conv.weight"], groups=h.shape[1])
    h = F.silu(_ln(sd, p + "depthwise_layer_norm", h.transpose(1, 2), cfg.layer_norm_eps)).transpose(1, 2)
    return F.conv1d(h, sd[p + "pointwise_conv2.weight"]).transpose(1, 2)


def hidden_states(sd, cfg: W2VBertCfg, input_features: torch.Tensor, attention_mask: torch.Tensor = None, n_layers: int = None) -> List[torch.Tensor]:
    """Wav2Vec2BertModel(..., output_hidden_states=True).hidden_states[0 .. n_layers]: input_features (B, T, 160), attention_mask (B, T)."""
    x = F.linear(_ln(sd, "feature_projection.layer_norm", input_features, cfg.layer_norm_eps), sd["feature_projection.projection.weight"],
                 sd["feature_projection.projection.bias"])
    add_mask = None
    if attention_mask is not None:
        x = x.masked_fill(~attention_mask.bool().unsqueeze(-1), 0.0)
        add_mask = ((1.0 - attention_mask[:, None, None, :].float()) * torch.finfo(x.dtype).min).expand(-1, 1, attention_mask.shape[-1], -1)
    outs = [x]
    for i in range(cfg.num_hidden_layers if n_layers is None else n_layers):
        p = f"encoder.layers.{i}."
        x = _ffn(sd, p + "ffn1", _ln(sd, p + "ffn1_layer_norm", x, cfg.layer_norm_eps)) * 0.5 + x
        x = _self_attn(sd, p + "self_attn.", cfg, _ln(sd, p + "self_attn_layer_norm", x, cfg.layer_norm_eps), add_mask) + x
        x = x + _conv_module(sd, p + "conv_module.", cfg, x, attention_mask)
        x = _ffn(sd, p + "ffn2", _ln(sd, p + "ffn2_layer_norm", x, cfg.layer_norm_eps)) * 0.5 + x
        x = _ln(sd, p + "final_layer_norm", x, cfg.layer_norm_eps)
        outs.append(x)
    return outs


def get_emb(sd, cfg, input_features, attention_mask, mean, std, layer: int = 17):                  # infer_v2_5.py:282-290
    return (hidden_states(sd, cfg, input_features, attention_mask, n_layers=layer)[layer] - mean) / std
