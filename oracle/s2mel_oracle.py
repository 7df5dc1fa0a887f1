"""CPU fp32 restatement of the reference's s2mel flow-matching decoder (the first "next" row):
CFM Euler solver with classifier-free guidance around the DiT estimator (gpt-fast transformer with adaptive RMSNorm, RoPE,
SwiGLU, U-ViT skip connections, WaveNet head).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and tools/make_golden_s2mel.py).  No product code may import this module.

Restates (reference file:line):
  indextts/s2mel/modules/flow_matching.py:30-115      BASECFM.inference / solve_euler (CFG-batched estimator call)
  indextts/s2mel/modules/diffusion_transformer.py:20-60,85-101,186-257   TimestepEmbedder, FinalLayer, DiT.forward
  indextts/s2mel/modules/gpt_fast/model.py:20-38,121-360   AdaptiveLayerNorm, Transformer, TransformerBlock, Attention,
                                                             FeedForward, RMSNorm, precompute_freqs_cis, apply_rotary_emb
  indextts/s2mel/modules/wavenet.py:112-174            WN (gated dilated convs, global conditioning)
  indextts/s2mel/modules/encodec.py:71-113,192-228     SConv1d reflect padding
  indextts/s2mel/modules/commons.py:133-141,155-159    fused_add_tanh_sigmoid_multiply, sequence_mask

PINNING: PINNED by running the reference's own `CFM` class (indextts/s2mel/modules/flow_matching.py, imported here with
torchaudio / librosa / munch stubbed, tools/ref_shim_s2mel.py) on the weights produced by `synth_weights` below; fixture
tests/golden/s2mel_cfm.npz, generating script tools/make_golden_s2mel.py.  The real model sizes (checkpoints/config.yaml)
are not in the repository: the fixture uses a small configuration that exercises every branch the inference path takes
(U-ViT skips, long skip, style condition, WaveNet head, CFG, prompt masking, ragged length).
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass
class S2MelConfig:
    hidden_dim: int = 512
    num_heads: int = 8
    depth: int = 13
    in_channels: int = 80
    content_dim: int = 512
    style_dim: int = 192
    wavenet_hidden: int = 512
    wavenet_layers: int = 8
    wavenet_kernel: int = 5
    wavenet_dilation_rate: int = 1
    content_codebook_size: int = 1024
    norm_eps: float = 1e-5
    rope_base: float = 10000.0
    block_size: int = 16384

    @property
    def head_dim(self) -> int:
        return self.hidden_dim // self.num_heads

    @property
    def intermediate_size(self) -> int:            # gpt_fast/model.py:62-65
        n_hidden = int(2 * (4 * self.hidden_dim) / 3)
        return n_hidden if n_hidden % 256 == 0 else n_hidden + 256 - (n_hidden % 256)


# ----------------------------------------------------------------------------
# parameter inventory (reference state-dict names, in the reference's order) and seeded weights
# ----------------------------------------------------------------------------
def param_shapes(c: S2MelConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    H, I, W = c.hidden_dim, c.intermediate_size, c.wavenet_hidden
    out: List[Tuple[str, Tuple[int, ...]]] = []
    P = "estimator."
    for i in range(c.depth):
        L = f"{P}transformer.layers.{i}."
        out += [(L + "attention.wqkv.weight", (3 * H, H)), (L + "attention.wo.weight", (H, H)),
                (L + "feed_forward.w1.weight", (I, H)), (L + "feed_forward.w3.weight", (I, H)),
                (L + "feed_forward.w2.weight", (H, I)),
                (L + "ffn_norm.project_layer.weight", (2 * H, H)), (L + "ffn_norm.project_layer.bias", (2 * H,)),
                (L + "ffn_norm.norm.weight", (H,)),
                (L + "attention_norm.project_layer.weight", (2 * H, H)), (L + "attention_norm.project_layer.bias", (2 * H,)),
                (L + "attention_norm.norm.weight", (H,)),
                (L + "skip_in_linear.weight", (H, 2 * H)), (L + "skip_in_linear.bias", (H,))]
    out += [(P + "transformer.norm.project_layer.weight", (2 * H, H)), (P + "transformer.norm.project_layer.bias", (2 * H,)),
            (P + "transformer.norm.norm.weight", (H,)),
            (P + "x_embedder.bias", (H,)), (P + "x_embedder.weight_g", (H, 1)), (P + "x_embedder.weight_v", (H, c.in_channels)),
            (P + "cond_embedder.weight", (c.content_codebook_size, H)),
            (P + "cond_projection.weight", (H, c.content_dim)), (P + "cond_projection.bias", (H,)),
            (P + "t_embedder.freqs", (128,)),
            (P + "t_embedder.mlp.0.weight", (H, 256)), (P + "t_embedder.mlp.0.bias", (H,)),
            (P + "t_embedder.mlp.2.weight", (H, H)), (P + "t_embedder.mlp.2.bias", (H,)),
            (P + "t_embedder2.freqs", (128,)),
            (P + "t_embedder2.mlp.0.weight", (W, 256)), (P + "t_embedder2.mlp.0.bias", (W,)),
            (P + "t_embedder2.mlp.2.weight", (W, W)), (P + "t_embedder2.mlp.2.bias", (W,)),
            (P + "conv1.weight", (W, H)), (P + "conv1.bias", (W,)),
            (P + "conv2.weight", (c.in_channels, W, 1)), (P + "conv2.bias", (c.in_channels,))]
    for i in range(c.wavenet_layers):
        out += [(f"{P}wavenet.in_layers.{i}.conv.conv.bias", (2 * W,)), (f"{P}wavenet.in_layers.{i}.conv.conv.weight_g", (2 * W, 1, 1)),
                (f"{P}wavenet.in_layers.{i}.conv.conv.weight_v", (2 * W, W, c.wavenet_kernel))]
    for i in range(c.wavenet_layers):
        ro = 2 * W if i < c.wavenet_layers - 1 else W
        out += [(f"{P}wavenet.res_skip_layers.{i}.conv.conv.bias", (ro,)), (f"{P}wavenet.res_skip_layers.{i}.conv.conv.weight_g", (ro, 1, 1)),
                (f"{P}wavenet.res_skip_layers.{i}.conv.conv.weight_v", (ro, W, 1))]
    G = 2 * W * c.wavenet_layers
    out += [(P + "wavenet.cond_layer.conv.conv.bias", (G,)), (P + "wavenet.cond_layer.conv.conv.weight_g", (G, 1, 1)),
            (P + "wavenet.cond_layer.conv.conv.weight_v", (G, W, 1)),
            (P + "final_layer.linear.bias", (W,)), (P + "final_layer.linear.weight_g", (W, 1)), (P + "final_layer.linear.weight_v", (W, W)),
            (P + "final_layer.adaLN_modulation.1.weight", (2 * W, W)), (P + "final_layer.adaLN_modulation.1.bias", (2 * W,)),
            (P + "res_projection.weight", (W, H)), (P + "res_projection.bias", (W,)),
            (P + "content_mask_embedder.weight", (1, H)),
            (P + "skip_linear.weight", (H, H + c.in_channels)), (P + "skip_linear.bias", (H,)),
            (P + "cond_x_merge_linear.weight", (H, H + 2 * c.in_channels + c.style_dim)), (P + "cond_x_merge_linear.bias", (H,))]
    return out


def synth_weights(c: S2MelConfig, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Seeded, roughly variance-preserving weights under the reference's names (strict-loadable into the reference CFM)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(c):
        if name.endswith(".freqs"):
            half = shape[0]
            sd[name] = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
        elif name.endswith("norm.weight"):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("project_layer.bias") or name.endswith("adaLN_modulation.1.bias"):
            b = 0.05 * torch.randn(shape, generator=g)
            if name.endswith("project_layer.bias"):
                b[: shape[0] // 2] += 1.0                     # the "weight" half of the adaptive norm
            sd[name] = b
        elif name.endswith(".bias"):
            sd[name] = 0.02 * torch.randn(shape, generator=g)
        elif name.endswith("weight_g"):
            sd[name] = None                                    # filled from weight_v below
        elif "cond_embedder.weight" in name or "content_mask_embedder.weight" in name:
            sd[name] = torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            scale = 0.5 if ("project_layer" in name or "adaLN" in name) else 1.0
            sd[name] = torch.randn(shape, generator=g) * (scale / math.sqrt(fan_in))
    for name in list(sd):
        if name.endswith("weight_g"):
            v = sd[name[:-1] + "v"]
            norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(sd_shape(name, c))
            sd[name] = norm * (1.0 + 0.1 * torch.randn(norm.shape, generator=g))
    return sd


def sd_shape(name: str, c: S2MelConfig) -> Tuple[int, ...]:
    return dict(param_shapes(c))[name]


def _wn(sd, prefix: str) -> torch.Tensor:
    """weight-norm fold: w = g * v / ||v|| (norm over all dims but 0), torch.nn.utils.weight_norm semantics"""
    v, g = sd[prefix + "weight_v"], sd[prefix + "weight_g"]
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
    return v * (g / n)


# ----------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------
def timestep_embed(sd, prefix: str, t: torch.Tensor) -> torch.Tensor:
    """TimestepEmbedder (diffusion_transformer.py:20-60): scale 1000, 256 sinusoid features (cos | sin), MLP with SiLU."""
    args = 1000 * t[:, None].float() * sd[prefix + "freqs"][None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    h = F.silu(F.linear(emb, sd[prefix + "mlp.0.weight"], sd[prefix + "mlp.0.bias"]))
    return F.linear(h, sd[prefix + "mlp.2.weight"], sd[prefix + "mlp.2.bias"])


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    return x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps) * w


def ada_norm(sd, prefix: str, x: torch.Tensor, cemb: torch.Tensor, eps: float) -> torch.Tensor:
    """AdaptiveLayerNorm (gpt_fast/model.py:20-38): (weight, bias) = split(project_layer(c)); weight * RMSNorm(x) + bias."""
    wb = F.linear(cemb, sd[prefix + "project_layer.weight"], sd[prefix + "project_layer.bias"])
    w, b = torch.split(wb, x.shape[-1], dim=-1)
    return w * rms_norm(x, sd[prefix + "norm.weight"], eps) + b


def rope_table(c: S2MelConfig, T: int) -> torch.Tensor:
    """precompute_freqs_cis (gpt_fast/model.py:336-345) rows 0..T-1: (T, head_dim/2, 2) = (cos, sin)."""
    n = c.head_dim
    freqs = 1.0 / (c.rope_base ** (torch.arange(0, n, 2)[: n // 2].float() / n))
    ang = torch.outer(torch.arange(T).float(), freqs)
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1)


def apply_rope(x: torch.Tensor, tab: torch.Tensor) -> torch.Tensor:
    """apply_rotary_emb (gpt_fast/model.py:348-360): interleaved pairs (2i, 2i+1).  x (B, T, H, hd)."""
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    f = tab.view(1, xs.size(1), 1, xs.size(3), 2)
    out = torch.stack([xs[..., 0] * f[..., 0] - xs[..., 1] * f[..., 1],
                       xs[..., 1] * f[..., 0] + xs[..., 0] * f[..., 1]], -1)
    return out.flatten(3)


# see oracle/bigvgan_oracle.py::TIMING_MODE: set by bench.py's cpu_baseline leg only; the checker keeps the written-out attention below
TIMING_MODE = False


def attention(sd, prefix: str, c: S2MelConfig, x: torch.Tensor, tab: torch.Tensor, key_mask: torch.Tensor) -> torch.Tensor:
    """Attention.forward (gpt_fast/model.py:262-307), n_local_heads == n_head, no KV cache; key_mask (B, T) True = attend."""
    B, T, _ = x.shape
    H, hd = c.num_heads, c.head_dim
    q, k, v = F.linear(x, sd[prefix + "wqkv.weight"]).split([H * hd] * 3, dim=-1)
    q = apply_rope(q.view(B, T, H, hd), tab).transpose(1, 2)
    k = apply_rope(k.view(B, T, H, hd), tab).transpose(1, 2)
    v = v.view(B, T, H, hd).transpose(1, 2)
    if TIMING_MODE:        # bench.py's cpu_baseline only: the fused CPU attention the reference calls (gpt_fast/model.py:303), not the checked form
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=key_mask[:, None, None, :].expand(B, 1, T, T))
        return F.linear(y.transpose(1, 2).reshape(B, T, H * hd), sd[prefix + "wo.weight"])
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    y = torch.softmax(s, dim=-1) @ v
    return F.linear(y.transpose(1, 2).reshape(B, T, H * hd), sd[prefix + "wo.weight"])


def transformer(sd, c: S2MelConfig, x: torch.Tensor, cemb: torch.Tensor, key_mask: torch.Tensor) -> torch.Tensor:
    """Transformer.forward (gpt_fast/model.py:161-193) with U-ViT skips: layers i < n//2 emit, layers i > n//2 receive."""
    P = "estimator.transformer."
    tab = rope_table(c, x.shape[1])
    skips: List[torch.Tensor] = []
    n = c.depth
    for i in range(n):
        L = f"{P}layers.{i}."
        if i > n // 2:
            x = F.linear(torch.cat([x, skips.pop(-1)], dim=-1), sd[L + "skip_in_linear.weight"], sd[L + "skip_in_linear.bias"])
        h = x + attention(sd, L + "attention.", c, ada_norm(sd, L + "attention_norm.", x, cemb, c.norm_eps), tab, key_mask)
        z = ada_norm(sd, L + "ffn_norm.", h, cemb, c.norm_eps)
        ff = F.linear(F.silu(F.linear(z, sd[L + "feed_forward.w1.weight"])) * F.linear(z, sd[L + "feed_forward.w3.weight"]),
                      sd[L + "feed_forward.w2.weight"])
        x = h + ff
        if i < n // 2:
            skips.append(x)
    return ada_norm(sd, P + "norm.", x, cemb, c.norm_eps)


def sconv1d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, dilation: int) -> torch.Tensor:
    """SConv1d.forward, stride 1, non-causal (encodec.py:212-228): reflect padding of (k-1)*d, left half rounded up;
    inputs not longer than the pad are zero-extended on the right first (pad1d, :96-113)."""
    k_eff = (w.shape[-1] - 1) * dilation + 1
    total = k_eff - 1
    right = total // 2
    left = total - right
    if total > 0:
        length = x.shape[-1]
        extra = 0
        if length <= max(left, right):
            extra = max(left, right) - length + 1
            x = F.pad(x, (0, extra))
        x = F.pad(x, (left, right), mode="reflect")
        x = x[..., : x.shape[-1] - extra]
    return F.conv1d(x, w, b, dilation=dilation)


def wavenet(sd, c: S2MelConfig, x: torch.Tensor, x_mask: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """WN.forward (wavenet.py:143-166).  x (B, W, T), x_mask (B, 1, T) bool, g (B, W, 1)."""
    P = "estimator.wavenet."
    W = c.wavenet_hidden
    out = torch.zeros_like(x)
    gc = F.conv1d(g, _wn(sd, P + "cond_layer.conv.conv."), sd[P + "cond_layer.conv.conv.bias"])
    for i in range(c.wavenet_layers):
        d = c.wavenet_dilation_rate ** i
        x_in = sconv1d(x, _wn(sd, f"{P}in_layers.{i}.conv.conv."), sd[f"{P}in_layers.{i}.conv.conv.bias"], d)
        a = x_in + gc[:, i * 2 * W:(i + 1) * 2 * W, :]
        acts = torch.tanh(a[:, :W]) * torch.sigmoid(a[:, W:])
        rs = F.conv1d(acts, _wn(sd, f"{P}res_skip_layers.{i}.conv.conv."), sd[f"{P}res_skip_layers.{i}.conv.conv.bias"])
        if i < c.wavenet_layers - 1:
            x = (x + rs[:, :W]) * x_mask
            out = out + rs[:, W:]
        else:
            out = out + rs
    return out * x_mask


# ----------------------------------------------------------------------------
# DiT estimator and the CFM solver
# ----------------------------------------------------------------------------
def dit_forward(sd, c: S2MelConfig, x: torch.Tensor, prompt_x: torch.Tensor, x_lens: torch.Tensor, t: torch.Tensor,
                style: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
    """DiT.forward in eval mode (diffusion_transformer.py:186-257): style_condition, long_skip_connection, wavenet head,
    non-causal; x, prompt_x (B, 80, T); t (B,); style (B, style_dim); cond (B, T, content_dim) -> (B, 80, T)."""
    P = "estimator."
    B, _, T = x.shape
    t1 = timestep_embed(sd, P + "t_embedder.", t)
    cond = F.linear(cond, sd[P + "cond_projection.weight"], sd[P + "cond_projection.bias"])
    xt = x.transpose(1, 2)
    x_in = torch.cat([xt, prompt_x.transpose(1, 2), cond, style[:, None, :].repeat(1, T, 1)], dim=-1)
    x_in = F.linear(x_in, sd[P + "cond_x_merge_linear.weight"], sd[P + "cond_x_merge_linear.bias"])
    key_mask = torch.arange(T)[None, :] < x_lens[:, None]                  # sequence_mask, broadcast over the CFG batch
    if key_mask.shape[0] != B:
        key_mask = key_mask.expand(B, -1)
    x_res = transformer(sd, c, x_in, t1.unsqueeze(1), key_mask)
    x_res = F.linear(torch.cat([x_res, xt], dim=-1), sd[P + "skip_linear.weight"], sd[P + "skip_linear.bias"])
    h = F.linear(x_res, sd[P + "conv1.weight"], sd[P + "conv1.bias"]).transpose(1, 2)
    t2 = timestep_embed(sd, P + "t_embedder2.", t)
    h = wavenet(sd, c, h, key_mask[:, None, :], t2.unsqueeze(2)).transpose(1, 2)
    h = h + F.linear(x_res, sd[P + "res_projection.weight"], sd[P + "res_projection.bias"])
    # FinalLayer (:85-101): LayerNorm without affine (eps 1e-6), adaLN shift/scale from SiLU(t1), weight-normed linear
    mod = F.linear(F.silu(t1), sd[P + "final_layer.adaLN_modulation.1.weight"], sd[P + "final_layer.adaLN_modulation.1.bias"])
    shift, scale = mod.chunk(2, dim=1)
    h = F.layer_norm(h, (h.shape[-1],), None, None, 1e-6) * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)
    h = F.linear(h, _wn(sd, P + "final_layer.linear."), sd[P + "final_layer.linear.bias"]).transpose(1, 2)
    return F.conv1d(h, sd[P + "conv2.weight"], sd[P + "conv2.bias"])


def cfm_solve_euler(sd, c: S2MelConfig, z: torch.Tensor, x_lens: torch.Tensor, prompt: torch.Tensor, mu: torch.Tensor,
                    style: torch.Tensor, n_timesteps: int, inference_cfg_rate: float = 0.7) -> torch.Tensor:
    """BASECFM.inference / solve_euler (flow_matching.py:30-115) from a given noise z (B=1, 80, T): fixed-step Euler over
    t in linspace(0, 1, n+1); each step one estimator call on the CFG-stacked batch [cond ; null]; the prompt frames of x
    are held at 0 and fed through prompt_x."""
    x = z.clone()
    t_span = torch.linspace(0, 1, n_timesteps + 1)
    t = t_span[0]
    prompt_len = prompt.size(-1)
    prompt_x = torch.zeros_like(x)
    prompt_x[..., :prompt_len] = prompt[..., :prompt_len]
    x[..., :prompt_len] = 0
    for step in range(1, len(t_span)):
        dt = t_span[step] - t_span[step - 1]
        if inference_cfg_rate > 0:
            d = dit_forward(sd, c, torch.cat([x, x], 0), torch.cat([prompt_x, torch.zeros_like(prompt_x)], 0), x_lens,
                            torch.stack([t, t]), torch.cat([style, torch.zeros_like(style)], 0),
                            torch.cat([mu, torch.zeros_like(mu)], 0))
            dphi, cfg_dphi = d.chunk(2, dim=0)
            dphi = (1.0 + inference_cfg_rate) * dphi - inference_cfg_rate * cfg_dphi
        else:
            dphi = dit_forward(sd, c, x, prompt_x, x_lens, t.unsqueeze(0), style, mu)
        x = x + dt * dphi
        t = t + dt
        x[:, :, :prompt_len] = 0
    return x
