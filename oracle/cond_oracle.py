"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32) of the conditioning encoders of IndexTTS-2 / 2.5:
the Conformer encoder + Perceiver resampler pairs behind `UnifiedVoice.get_conditioning` / `get_emo_conditioning` / `get_emovec` /
`merge_emovec`.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows (paths relative to the reference repo root):
  ConformerEncoder / BaseEncoder.forward      indextts/gpt/conformer_encoder.py:316-520 (forward :398-433)
  ConformerEncoderLayer.forward               indextts/gpt/conformer_encoder.py:247-313 (normalize_before, no macaron, conv module)
  ConvolutionModule.forward                   indextts/gpt/conformer_encoder.py:112-160 (pointwise -> GLU -> depthwise k -> LayerNorm -> SiLU -> pointwise)
  PositionwiseFeedForward (SiLU)              indextts/gpt/conformer_encoder.py:20-53
  RelPositionMultiHeadedAttention.forward     indextts/gpt/conformer/attention.py:232-312 (+ forward_attention :85-120)
  Conv2dSubsampling2                          indextts/gpt/conformer/subsampling.py:135-186
  RelPositionalEncoding / PositionalEncoding  indextts/gpt/conformer/embedding.py:20-141
  make_pad_mask                               indextts/utils/common.py:135-158
  PerceiverResampler / Attention / Attend / FeedForward (GEGLU) / RMSNorm     indextts/gpt/perceiver.py
  get_conditioning / get_emo_conditioning / get_emovec / merge_emovec          indextts/gpt/model_v2.py:556-593,827-838

Pinned by tests/golden/cond.npz: outputs of the reference's own classes on this module's seeded weights (tools/make_golden_cond.py).
"""
import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass
class ConformerCfg:
    input_size: int = 1024
    output_size: int = 512
    attention_heads: int = 8
    linear_units: int = 2048
    num_blocks: int = 6
    cnn_kernel: int = 15


@dataclass
class PerceiverCfg:
    dim: int = 1280
    dim_context: int = 512
    num_latents: int = 32
    dim_head: int = 64
    heads: int = 8
    ff_mult: float = 2.0
    depth: int = 2


def synth_conformer(cfg: ConformerCfg, seed: int, pre: str = "") -> Dict[str, torch.Tensor]:
    """Seeded weights under the reference ConformerEncoder's parameter names."""
    g = torch.Generator().manual_seed(seed)
    D, H, U, k = cfg.output_size, cfg.attention_heads, cfg.linear_units, cfg.cnn_kernel
    f_out = (cfg.input_size - 1) // 2
    rn = lambda *s, fan: torch.randn(*s, generator=g) / math.sqrt(fan)
    sd = {pre + "embed.conv.0.weight": rn(D, 1, 3, 3, fan=9), pre + "embed.conv.0.bias": 0.1 * torch.randn(D, generator=g),
          pre + "embed.out.0.weight": rn(D, D * f_out, fan=D * f_out) * 2.0, pre + "embed.out.0.bias": 0.05 * torch.randn(D, generator=g),
          pre + "after_norm.weight": 1 + 0.1 * torch.randn(D, generator=g), pre + "after_norm.bias": 0.05 * torch.randn(D, generator=g)}
    for i in range(cfg.num_blocks):
        p = f"{pre}encoders.{i}."
        for n in ("q", "k", "v", "out"):
            sd[p + f"self_attn.linear_{n}.weight"] = rn(D, D, fan=D)
            sd[p + f"self_attn.linear_{n}.bias"] = 0.05 * torch.randn(D, generator=g)
        sd[p + "self_attn.linear_pos.weight"] = rn(D, D, fan=D)
        sd[p + "self_attn.pos_bias_u"] = 0.3 * torch.randn(H, D // H, generator=g)
        sd[p + "self_attn.pos_bias_v"] = 0.3 * torch.randn(H, D // H, generator=g)
        sd[p + "feed_forward.w_1.weight"], sd[p + "feed_forward.w_1.bias"] = rn(U, D, fan=D), 0.05 * torch.randn(U, generator=g)
        sd[p + "feed_forward.w_2.weight"], sd[p + "feed_forward.w_2.bias"] = rn(D, U, fan=U), 0.05 * torch.randn(D, generator=g)
        sd[p + "conv_module.pointwise_conv1.weight"] = rn(2 * D, D, 1, fan=D)
        sd[p + "conv_module.pointwise_conv1.bias"] = 0.05 * torch.randn(2 * D, generator=g)
        sd[p + "conv_module.depthwise_conv.weight"] = rn(D, 1, k, fan=k) * 1.5
        sd[p + "conv_module.depthwise_conv.bias"] = 0.05 * torch.randn(D, generator=g)
        sd[p + "conv_module.norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
        sd[p + "conv_module.norm.bias"] = 0.05 * torch.randn(D, generator=g)
        sd[p + "conv_module.pointwise_conv2.weight"] = rn(D, D, 1, fan=D)
        sd[p + "conv_module.pointwise_conv2.bias"] = 0.05 * torch.randn(D, generator=g)
        for n in ("norm_ff", "norm_mha", "norm_conv", "norm_final"):
            sd[p + n + ".weight"] = 1 + 0.1 * torch.randn(D, generator=g)
            sd[p + n + ".bias"] = 0.05 * torch.randn(D, generator=g)
    return sd


def synth_perceiver(cfg: PerceiverCfg, seed: int, pre: str = "") -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    D, Dc, inner = cfg.dim, cfg.dim_context, cfg.dim_head * cfg.heads
    ffi = int(cfg.dim * cfg.ff_mult * 2 / 3)
    rn = lambda *s, fan: torch.randn(*s, generator=g) / math.sqrt(fan)
    sd = {pre + "latents": 0.5 * torch.randn(cfg.num_latents, D, generator=g), pre + "norm.gamma": 1 + 0.1 * torch.randn(D, generator=g)}
    if Dc != D:
        sd[pre + "proj_context.weight"], sd[pre + "proj_context.bias"] = rn(D, Dc, fan=Dc), 0.05 * torch.randn(D, generator=g)
    for i in range(cfg.depth):
        p = f"{pre}layers.{i}."
        sd[p + "0.to_q.weight"] = rn(inner, D, fan=D)
        sd[p + "0.to_kv.weight"] = rn(2 * inner, D, fan=D)
        sd[p + "0.to_out.weight"] = rn(D, inner, fan=inner)
        sd[p + "1.0.weight"], sd[p + "1.0.bias"] = rn(2 * ffi, D, fan=D), 0.05 * torch.randn(2 * ffi, generator=g)
        sd[p + "1.2.weight"], sd[p + "1.2.bias"] = rn(D, ffi, fan=ffi), 0.05 * torch.randn(D, generator=g)
    return sd


def make_pad_mask(lengths: torch.Tensor, max_len: int) -> torch.Tensor:          # utils/common.py:135-158
    return torch.arange(max_len)[None, :] >= lengths.reshape(-1, 1)


def positional_table(d_model: int, n: int) -> torch.Tensor:                      # embedding.py:36-44
    pe = torch.zeros(n, d_model)
    position = torch.arange(0, n).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def rel_pos_attention(sd, p, H, x, mask, pos_emb):                                # attention.py:232-312, 85-120
    B, T, D = x.shape
    dk = D // H
    lin = lambda n, t: F.linear(t, sd[p + f"linear_{n}.weight"], sd.get(p + f"linear_{n}.bias"))
    q = lin("q", x).view(B, T, H, dk)
    k = lin("k", x).view(B, T, H, dk).transpose(1, 2)
    v = lin("v", x).view(B, T, H, dk).transpose(1, 2)
    pp = F.linear(pos_emb, sd[p + "linear_pos.weight"]).view(1, -1, H, dk).transpose(1, 2)
    qu = (q + sd[p + "pos_bias_u"]).transpose(1, 2)
    qv = (q + sd[p + "pos_bias_v"]).transpose(1, 2)
    scores = (qu @ k.transpose(-2, -1) + qv @ pp.transpose(-2, -1)) / math.sqrt(dk)
    m = mask.unsqueeze(1).eq(0)                                                   # (B,1,1,T)
    attn = torch.softmax(scores.masked_fill(m, -float("inf")), dim=-1).masked_fill(m, 0.0)
    out = (attn @ v).transpose(1, 2).contiguous().view(B, T, D)
    return lin("out", out)


def conv_module(sd, p, x, mask_pad, k):                                           # conformer_encoder.py:112-160
    x = x.transpose(1, 2).masked_fill(~mask_pad, 0.0)
    x = F.glu(F.conv1d(x, sd[p + "pointwise_conv1.weight"], sd[p + "pointwise_conv1.bias"]), dim=1)
    x = F.conv1d(x, sd[p + "depthwise_conv.weight"], sd[p + "depthwise_conv.bias"], padding=(k - 1) // 2, groups=x.shape[1])
    x = F.silu(_ln(sd, p + "norm", x.transpose(1, 2))).transpose(1, 2)
    x = F.conv1d(x, sd[p + "pointwise_conv2.weight"], sd[p + "pointwise_conv2.bias"])
    return x.masked_fill(~mask_pad, 0.0).transpose(1, 2)


def conformer_encoder(sd, cfg: ConformerCfg, xs: torch.Tensor, xs_lens: torch.Tensor, pre: str = "") -> Tuple[torch.Tensor, torch.Tensor]:
    """xs (B, T, input_size), xs_lens (B,) -> (B, T', output_size), mask (B, 1, T'), T' = (T - 1) // 2."""
    T = xs.shape[1]
    masks = ~make_pad_mask(xs_lens, T).unsqueeze(1)
    x = F.relu(F.conv2d(xs.unsqueeze(1), sd[pre + "embed.conv.0.weight"], sd[pre + "embed.conv.0.bias"], stride=2))
    b, c, t, f = x.shape
    x = F.linear(x.transpose(1, 2).contiguous().view(b, t, c * f), sd[pre + "embed.out.0.weight"], sd[pre + "embed.out.0.bias"])
    x = x * math.sqrt(cfg.output_size)
    pos_emb = positional_table(cfg.output_size, t).unsqueeze(0)
    masks = masks[:, :, 2::2]
    for i in range(cfg.num_blocks):
        p = f"{pre}encoders.{i}."
        x = x + rel_pos_attention(sd, p + "self_attn.", cfg.attention_heads, _ln(sd, p + "norm_mha", x), masks, pos_emb)
        x = x + conv_module(sd, p + "conv_module.", _ln(sd, p + "norm_conv", x), masks, cfg.cnn_kernel)
        h = _ln(sd, p + "norm_ff", x)
        x = x + F.linear(F.silu(F.linear(h, sd[p + "feed_forward.w_1.weight"], sd[p + "feed_forward.w_1.bias"])),
                         sd[p + "feed_forward.w_2.weight"], sd[p + "feed_forward.w_2.bias"])
        x = _ln(sd, p + "norm_final", x)
    return _ln(sd, pre + "after_norm", x), masks


def perceiver_resampler(sd, cfg: PerceiverCfg, x: torch.Tensor, mask: torch.Tensor, pre: str = "") -> torch.Tensor:
    """x (B, T, dim_context), mask (B, num_latents + T) bool (True = attend) -> (B, num_latents, dim)   (perceiver.py)"""
    B = x.shape[0]
    if pre + "proj_context.weight" in sd:
        x = F.linear(x, sd[pre + "proj_context.weight"], sd[pre + "proj_context.bias"])
    lat = sd[pre + "latents"].unsqueeze(0).expand(B, -1, -1)
    H, dh = cfg.heads, cfg.dim_head
    for i in range(cfg.depth):
        p = f"{pre}layers.{i}."
        ctx = torch.cat((lat, x), dim=-2)                                        # cross_attn_include_queries
        q = F.linear(lat, sd[p + "0.to_q.weight"]).view(B, -1, H, dh).transpose(1, 2)
        kv = F.linear(ctx, sd[p + "0.to_kv.weight"])
        k, v = (t.view(B, -1, H, dh).transpose(1, 2) for t in kv.chunk(2, dim=-1))
        sim = (q @ k.transpose(-2, -1)) * dh ** -0.5
        sim = sim.masked_fill(~mask[:, None, None, :], -torch.finfo(sim.dtype).max)
        out = (sim.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, -1, H * dh)
        lat = F.linear(out, sd[p + "0.to_out.weight"]) + lat
        h = F.linear(lat, sd[p + "1.0.weight"], sd[p + "1.0.bias"])
        a, gate = h.chunk(2, dim=-1)
        lat = F.linear(F.gelu(gate) * a, sd[p + "1.2.weight"], sd[p + "1.2.bias"]) + lat
    return F.normalize(lat, dim=-1) * cfg.dim ** 0.5 * sd[pre + "norm.gamma"]


def conditioning(sd, ccfg: ConformerCfg, pcfg: PerceiverCfg, feats: torch.Tensor, lens: torch.Tensor, enc_pre: str, perc_pre: str) -> torch.Tensor:
    """`get_conditioning` (conformer_perceiver) / `get_emo_conditioning` (model_v2.py:563-568, 588-593): feats (B, T, 1024) ->
    (B, num_latents, dim).  The perceiver mask is the encoder's key mask left-padded with True for the latents (cond_mask_pad)."""
    h, mask = conformer_encoder(sd, ccfg, feats, lens, enc_pre)
    m = F.pad(mask.squeeze(1), (pcfg.num_latents, 0), value=True)
    return perceiver_resampler(sd, pcfg, h, m, perc_pre)


def get_emovec(sd, ccfg, pcfg, feats, lens):                                     # model_v2.py:827-831
    v = conditioning(sd, ccfg, pcfg, feats, lens, "emo_conditioning_encoder.", "emo_perceiver_encoder.").squeeze(1)
    v = F.linear(v, sd["emovec_layer.weight"], sd["emovec_layer.bias"])
    return F.linear(v, sd["emo_layer.weight"], sd["emo_layer.bias"])


def merge_emovec(sd, ccfg, pcfg, spk_feats, emo_feats, spk_lens, emo_lens, alpha: float = 1.0):      # model_v2.py:833-838
    emo, base = get_emovec(sd, ccfg, pcfg, emo_feats, emo_lens), get_emovec(sd, ccfg, pcfg, spk_feats, spk_lens)
    return base + alpha * (emo - base)
