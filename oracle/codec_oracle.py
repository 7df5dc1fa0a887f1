"""CPU fp32 restatement of the step between the GPT codes and the flow-matching decoder (the second "next" row, SURVEY.md
section 8f-2): `EnhancedCodec.decode` (codebook lookup, Vocos ConvNeXt decoder, nearest 2x upsampling + conv) and the
s2mel `InterpolateRegulator` (continuous input, nearest interpolation to the target mel length, conv/GroupNorm/Mish stack).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and tools/make_golden_codec.py).  No product code may import this module.

Restates (reference file:line):
  indextts/codec/models.py:205-231                              EnhancedCodec.decode
  indextts/codec/models.py:179-199                              EnhancedCodec.quantize (stride-2 down conv + GELU, Vocos encoder, FVQ search)
  indextts/codec/amphion_codec/quantize/factorized_vector_quantize.py:52-118   FVQ.forward / decode_latents (L2-normalised nearest code)
  indextts/codec/amphion_codec/quantize/residual_vq.py:144-152  ResidualVQ.vq2emb
  indextts/codec/amphion_codec/quantize/factorized_vector_quantize.py:99-127   FVQ decode_code / vq2emb (weight-normed 1x1 out_project)
  indextts/codec/kmeans/vocos.py:468-526,719-782                ConvNeXtBlock, VocosBackbone
  indextts/s2mel/modules/length_regulator.py:28-141             InterpolateRegulator (is_discrete=False, no f0, no VQ)
  call sites: indextts/infer_v2_5.py:830-838

PINNING: PINNED by running the reference's own `EnhancedCodec` and `InterpolateRegulator` classes (torchaudio / DAC stubs,
tools/ref_shim_s2mel.py) on the weights of `synth_codec_weights` / `synth_regulator_weights`; fixture tests/golden/codec.npz,
generating script tools/make_golden_codec.py; `codec_quantize` likewise by tests/golden/codec_quantize.npz
(tools/make_golden_codec_quantize.py).  Model sizes in the fixture are small; the real ones are constructor defaults
(8192 x 8 codebook, hidden 1024, Vocos 384 / 2048 x 12).
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass
class CodecConfig:
    codebook_size: int = 8192
    hidden_size: int = 1024
    codebook_dim: int = 8
    vocos_dim: int = 384
    vocos_intermediate_dim: int = 2048
    vocos_num_layers: int = 12


@dataclass
class RegulatorConfig:
    channels: int = 512
    in_channels: int = 1024
    n_layers: int = 4               # len(sampling_ratios)
    groups: int = 1
    codebook_size: int = 1024       # unused embedding table of the discrete variant (present in the state dict)


def codec_param_shapes(c: CodecConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """The tensors `decode` touches, reference names."""
    D, H, I = c.vocos_dim, c.hidden_size, c.vocos_intermediate_dim
    out = [("up.weight", (H, H, 3)), ("up.bias", (H,)),
           ("decoder.0.embed.weight", (D, H, 7)), ("decoder.0.embed.bias", (D,)),
           ("decoder.0.norm.weight", (D,)), ("decoder.0.norm.bias", (D,))]
    for i in range(c.vocos_num_layers):
        p = f"decoder.0.convnext.{i}."
        out += [(p + "gamma", (D,)), (p + "dwconv.weight", (D, 1, 7)), (p + "dwconv.bias", (D,)),
                (p + "norm.weight", (D,)), (p + "norm.bias", (D,)),
                (p + "pwconv1.weight", (I, D)), (p + "pwconv1.bias", (I,)),
                (p + "pwconv2.weight", (D, I)), (p + "pwconv2.bias", (D,))]
    out += [("decoder.0.final_layer_norm.weight", (D,)), ("decoder.0.final_layer_norm.bias", (D,)),
            ("decoder.1.weight", (H, D)), ("decoder.1.bias", (H,)),
            ("quantizer.quantizers.0.out_project.bias", (H,)), ("quantizer.quantizers.0.out_project.weight_g", (H, 1, 1)),
            ("quantizer.quantizers.0.out_project.weight_v", (H, c.codebook_dim, 1)),
            ("quantizer.quantizers.0.codebook.weight", (c.codebook_size, c.codebook_dim))]
    return out


def regulator_param_shapes(c: RegulatorConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    C = c.channels
    out = [("mask_token", (1, C))]
    for i in range(c.n_layers):
        out += [(f"model.{3 * i}.weight", (C, C, 3)), (f"model.{3 * i}.bias", (C,)),
                (f"model.{3 * i + 1}.weight", (C,)), (f"model.{3 * i + 1}.bias", (C,))]
    out += [(f"model.{3 * c.n_layers}.weight", (C, C, 1)), (f"model.{3 * c.n_layers}.bias", (C,)),
            ("embedding.weight", (c.codebook_size, C)),
            ("content_in_proj.weight", (C, c.in_channels)), ("content_in_proj.bias", (C,))]
    return out


def _synth(shapes, seed: int) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in shapes:
        if name.endswith("weight_g"):
            continue
        if name.endswith("gamma"):
            sd[name] = 0.3 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("norm.weight") or (name.endswith(".weight") and len(shape) == 1):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias") or name == "mask_token":
            sd[name] = 0.05 * torch.randn(shape, generator=g)
        elif "codebook" in name or name == "embedding.weight":
            sd[name] = torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            sd[name] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
    for name, shape in shapes:
        if name.endswith("weight_g"):
            v = sd[name[:-1] + "v"]
            sd[name] = v.reshape(v.shape[0], -1).norm(dim=1).reshape(shape) * (1.0 + 0.1 * torch.randn(shape, generator=g))
    return sd


def synth_codec_weights(c: CodecConfig, seed: int = 1234) -> Dict[str, torch.Tensor]:
    return _synth(codec_param_shapes(c), seed)


def synth_regulator_weights(c: RegulatorConfig, seed: int = 1234) -> Dict[str, torch.Tensor]:
    return _synth(regulator_param_shapes(c), seed)


def codec_decode(sd, c: CodecConfig, codes: torch.Tensor) -> torch.Tensor:
    """codes (B, T) int -> (B, 2T, hidden)  (models.py:205-231)."""
    P = "quantizer.quantizers.0."
    emb = sd[P + "codebook.weight"][codes.long()].transpose(1, 2)                      # (B, cd, T)  decode_code
    v, g = sd[P + "out_project.weight_v"], sd[P + "out_project.weight_g"]
    w = v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape))
    x = F.conv1d(emb, w, sd[P + "out_project.bias"])                                   # (B, H, T)
    # VocosBackbone (vocos.py:770-782)
    D = c.vocos_dim
    x = F.conv1d(x, sd["decoder.0.embed.weight"], sd["decoder.0.embed.bias"], padding=3)
    x = F.layer_norm(x.transpose(1, 2), (D,), sd["decoder.0.norm.weight"], sd["decoder.0.norm.bias"], 1e-6).transpose(1, 2)
    for i in range(c.vocos_num_layers):
        p = f"decoder.0.convnext.{i}."
        y = F.conv1d(x, sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=3, groups=D).transpose(1, 2)
        y = F.layer_norm(y, (D,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
        y = F.linear(F.gelu(F.linear(y, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"])), sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"])
        x = x + (sd[p + "gamma"] * y).transpose(1, 2)
    x = F.layer_norm(x.transpose(1, 2), (D,), sd["decoder.0.final_layer_norm.weight"], sd["decoder.0.final_layer_norm.bias"], 1e-6)
    x = F.linear(x, sd["decoder.1.weight"], sd["decoder.1.bias"])                     # (B, T, H)
    x = F.interpolate(x.transpose(1, 2), scale_factor=2, mode="nearest")
    return F.conv1d(x, sd["up.weight"], sd["up.bias"], padding=1).transpose(1, 2)


def codec_encoder_param_shapes(c: CodecConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """The tensors `quantize` touches beyond those of `decode` (codebook, out_project), reference names."""
    D, H, I = c.vocos_dim, c.hidden_size, c.vocos_intermediate_dim
    out = [("down.weight", (H, H, 3)), ("down.bias", (H,)),
           ("encoder.0.embed.weight", (D, H, 7)), ("encoder.0.embed.bias", (D,)),
           ("encoder.0.norm.weight", (D,)), ("encoder.0.norm.bias", (D,))]
    for i in range(c.vocos_num_layers):
        p = f"encoder.0.convnext.{i}."
        out += [(p + "gamma", (D,)), (p + "dwconv.weight", (D, 1, 7)), (p + "dwconv.bias", (D,)),
                (p + "norm.weight", (D,)), (p + "norm.bias", (D,)),
                (p + "pwconv1.weight", (I, D)), (p + "pwconv1.bias", (I,)),
                (p + "pwconv2.weight", (D, I)), (p + "pwconv2.bias", (D,))]
    out += [("encoder.0.final_layer_norm.weight", (D,)), ("encoder.0.final_layer_norm.bias", (D,)),
            ("encoder.1.weight", (H, D)), ("encoder.1.bias", (H,)),
            ("quantizer.quantizers.0.in_project.bias", (c.codebook_dim,)), ("quantizer.quantizers.0.in_project.weight_g", (c.codebook_dim, 1, 1)),
            ("quantizer.quantizers.0.in_project.weight_v", (c.codebook_dim, H, 1))]
    return out


def synth_codec_encoder_weights(c: CodecConfig, seed: int = 4321) -> Dict[str, torch.Tensor]:
    """Seeded encoder-half weights (their own generator: the decode-half fixtures keep their values)."""
    return _synth(codec_encoder_param_shapes(c), seed)


def _wn(sd, p):
    v, g = sd[p + "weight_v"], sd[p + "weight_g"]
    return v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape))


def _vocos(sd, c: CodecConfig, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """VocosBackbone + the Linear after it (vocos.py:770-782): x (B, H, T) -> (B, T, H)"""
    D = c.vocos_dim
    x = F.conv1d(x, sd[prefix + "0.embed.weight"], sd[prefix + "0.embed.bias"], padding=3)
    x = F.layer_norm(x.transpose(1, 2), (D,), sd[prefix + "0.norm.weight"], sd[prefix + "0.norm.bias"], 1e-6).transpose(1, 2)
    for i in range(c.vocos_num_layers):
        p = f"{prefix}0.convnext.{i}."
        y = F.conv1d(x, sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=3, groups=D).transpose(1, 2)
        y = F.layer_norm(y, (D,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
        y = F.linear(F.gelu(F.linear(y, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"])), sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"])
        x = x + (sd[p + "gamma"] * y).transpose(1, 2)
    x = F.layer_norm(x.transpose(1, 2), (D,), sd[prefix + "0.final_layer_norm.weight"], sd[prefix + "0.final_layer_norm.bias"], 1e-6)
    return F.linear(x, sd[prefix + "1.weight"], sd[prefix + "1.bias"])


def codec_quantize(sd, c: CodecConfig, x: torch.Tensor):
    """`EnhancedCodec.quantize` / `RepCodec.quantize` (indextts/codec/models.py:179-199; call site indextts/infer_v2.py:465), eval mode, one
    quantizer: x (B, T, hidden) -> (indices (B, T'), quantized (B, T', hidden), margin (B, T')), T' = (T - 1) // 2 + 1.
    FVQ.forward / decode_latents: factorized_vector_quantize.py:52-118.  `margin` = best minus second-best negative distance of the
    L2-normalised search: where it is below float32 rounding an engine may pick the neighbouring code."""
    P = "quantizer.quantizers.0."
    h = F.gelu(F.conv1d(x.transpose(1, 2), sd["down.weight"], sd["down.bias"], stride=2, padding=1))
    h = _vocos(sd, c, "encoder.", h)                                                     # (B, T', H)
    z_e = F.conv1d(h.transpose(1, 2), _wn(sd, P + "in_project."), sd[P + "in_project.bias"])          # (B, cd, T')
    B = z_e.shape[0]
    enc = F.normalize(z_e.transpose(1, 2).reshape(-1, z_e.shape[1]))
    cb = F.normalize(sd[P + "codebook.weight"])
    dist = enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cb.t() + cb.pow(2).sum(1, keepdim=True).t()
    top2 = (-dist).topk(2, dim=1)
    idx = (-dist).max(1)[1].reshape(B, -1)
    z_q = sd[P + "codebook.weight"][idx].transpose(1, 2)                                 # decode_code: the raw (un-normalised) entries
    z_q = z_e + (z_q - z_e)                                                              # the straight-through form of FVQ.forward, as computed
    q = F.conv1d(z_q, _wn(sd, P + "out_project."), sd[P + "out_project.bias"])
    return idx, q.transpose(1, 2), (top2.values[:, 0] - top2.values[:, 1]).reshape(B, -1)


def length_regulator(sd, c: RegulatorConfig, x: torch.Tensor, ylens: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """InterpolateRegulator.forward, continuous input, eval (length_regulator.py:90-141): x (B, T, in_channels), ylens (B,)
    -> (B, max(ylens), channels) masked beyond each row's ylen, and ylens."""
    x = F.linear(x, sd["content_in_proj.weight"], sd["content_in_proj.bias"])
    Tm = int(ylens.max())
    mask = (torch.arange(Tm)[None, :] < ylens[:, None]).unsqueeze(-1)
    x = F.interpolate(x.transpose(1, 2).contiguous(), size=Tm, mode="nearest")
    for i in range(c.n_layers):
        x = F.conv1d(x, sd[f"model.{3 * i}.weight"], sd[f"model.{3 * i}.bias"], padding=1)
        x = F.group_norm(x, c.groups, sd[f"model.{3 * i + 1}.weight"], sd[f"model.{3 * i + 1}.bias"], 1e-5)
        x = F.mish(x)
    x = F.conv1d(x, sd[f"model.{3 * c.n_layers}.weight"], sd[f"model.{3 * c.n_layers}.bias"])
    return x.transpose(1, 2).contiguous() * mask, ylens
