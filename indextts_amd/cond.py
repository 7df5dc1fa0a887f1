"""Conditioning encoders of IndexTTS-2 / 2.5 on the HIP engine (first part): host mirrors of the reference's
`ConformerEncoder` (indextts/gpt/conformer_encoder.py, input layer conv2d2, rel_pos attention, conv module, no macaron) and
`PerceiverResampler` (indextts/gpt/perceiver.py) with the reference's constructor arguments and parameter names.  They sit behind
`UnifiedVoice.get_conditioning` / `get_emo_conditioning` / `get_emovec` / `merge_emovec` (model_v2.py:556-593,827-838).

Once-per-speaker work, so every op is an exact-f32 unit op of the C ABI on PACKED rows (only the valid frames of each prompt exist
as rows: the reference's key masks / zeroed padding have nothing to act on; every prompt of a batch gets the result the reference
computes for it ALONE -- inside a padded reference batch a shorter prompt's last frames additionally see GLU(bias) of the padded
positions through the depthwise conv, conformer_encoder.py:131-148, a batch-composition dependence the pipeline never exercises
because it encodes one prompt per call): dense layers on `itts_gemm_forward`, LayerNorms on
`itts_layernorm_forward`, the depthwise conv on `itts_tok_dwconv_forward`, attention on `itts_attention_forward`, GLU / GEGLU /
ReLU / SiLU / the Perceiver's RMSNorm on `itts_tok_{glu,act,l2norm}_forward`.  torch only gathers / concatenates rows.

How the reference arithmetic maps:
  * Conv2dSubsampling2 (subsampling.py:135-186): the 3x3 stride-2 conv is a GEMM over gathered 9-element patches (K padded to 16),
    ReLU, then the big Linear whose input columns are permuted once at load time from (channel, freq) to the GEMM's (freq, channel)
    order; the sqrt(d_model) input scale of the positional encoding (embedding.py:139) is folded into that Linear.
  * RelPositionMultiHeadedAttention (attention.py:232-312): matrix_ac + matrix_bd = [q + u | q + v] . [k | p]^T, i.e. ordinary
    attention with a doubled head dim; the biases u, v are folded into two copies of the query projection (one GEMM emits
    [q + u | q + v] per head, k and v), p = linear_pos(pos_emb) is one GEMM per layer over the positions.
"""
import math
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib
from .codec import F32, _TokOps, _tables
from .gpt import layernorm, pack_gemm_weight


def _pos_table(d_model: int, n: int) -> torch.Tensor:                            # embedding.py:36-44
    pe = torch.zeros(n, d_model)
    position = torch.arange(0, n).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


class _Ops(_TokOps):
    def attention(self, q, k, v, kstart, klen, heads, dq, dv, scale):
        out = torch.empty(q.shape[0], heads * dv, dtype=torch.float32, device=self.device)
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_attention_forward(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(out), _lib.ptr(kstart), _lib.ptr(klen),
                                                     q.shape[0], heads, dq, dv, float(scale), self._st()), "itts_attention_forward")
        return out

    def glu(self, x, mode):
        out = torch.empty(x.shape[0], x.shape[1] // 2, dtype=torch.float32, device=self.device)
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_glu_forward(_lib.ptr(x), _lib.ptr(out), x.shape[0], x.shape[1] // 2, mode, self._st()), "itts_tok_glu_forward")
        return out

    def act_(self, x, mode):
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_act_forward(_lib.ptr(x), x.numel(), mode, self._st()), "itts_tok_act_forward")
        return x

    def l2norm_(self, x, gamma, scale):
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_l2norm_forward(_lib.ptr(x), _lib.ptr(gamma), x.shape[0], x.shape[1], float(scale), self._st()),
                       "itts_tok_l2norm_forward")
        return x

    def add_(self, x, y):
        if not hasattr(self, "_ones") or self._ones.numel() < x.shape[1]:
            self._ones = torch.ones(max(2048, x.shape[1]), dtype=torch.float32, device=self.device)
        return self.scale_residual_(x, y, self._ones)


class _Lin:
    def __init__(self, w: torch.Tensor, b: Optional[torch.Tensor], device):
        """w [out][in] (nn.Linear); the input width is zero-padded to a multiple of 16 (the f32 MFMA K step)"""
        w = w.detach().float().cpu()
        self.n_out, self.k = w.shape[0], (w.shape[1] + 15) // 16 * 16
        if self.k != w.shape[1]:
            w = torch.nn.functional.pad(w, (0, self.k - w.shape[1]))
        self.wp = pack_gemm_weight(w, F32, transposed=True).to(device)
        self.b = None if b is None else b.detach().float().to(device).contiguous()


class ConformerEncoder:
    """indextts/gpt/conformer_encoder.py:436-520 with the defaults model_v2.py:359-364 relies on (rel_pos, normalize_before, conv
    module kernel 15, SiLU, no macaron); `input_layer` must be "conv2d2"."""

    def __init__(self, input_size: int, output_size: int = 256, attention_heads: int = 4, linear_units: int = 2048, num_blocks: int = 6,
                 dropout_rate: float = 0.0, input_layer: str = "conv2d2", pos_enc_layer_type: str = "rel_pos", normalize_before: bool = True,
                 concat_after: bool = False, macaron_style: bool = False, use_cnn_module: bool = True, cnn_module_kernel: int = 15,
                 device="cuda:0"):
        if input_layer != "conv2d2" or pos_enc_layer_type != "rel_pos" or not normalize_before or concat_after or macaron_style or not use_cnn_module:
            raise NotImplementedError("ConformerEncoder (HIP engine): conv2d2 / rel_pos / normalize_before / conv module / no macaron only")
        if output_size % 64 or (output_size // attention_heads) % 4 or (2 * output_size // attention_heads) > 256:
            raise ValueError("ConformerEncoder (HIP engine): output_size % 64 == 0 and head_dim % 4 == 0 required")
        self.idim, self.D, self.H, self.U, self.nb, self.k = input_size, output_size, attention_heads, linear_units, num_blocks, cnn_module_kernel
        self.device = torch.device(device)
        self.ops = _Ops(self.device)
        self._loaded = False

    def output_size(self) -> int:
        return self.D

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        D, H, dev = self.D, self.H, self.device
        dk, F_out = D // H, (self.idim - 1) // 2
        g = lambda k: sd[k].detach().float().cpu()
        dv = lambda k: g(k).to(dev).contiguous()
        # subsampling conv as a [9 -> D] GEMM (patch order i * 3 + j), then the Linear on (freq, channel)-ordered columns, x sqrt(D)
        self.sub_conv = _Lin(g("embed.conv.0.weight").reshape(D, 9), g("embed.conv.0.bias"), dev)
        w_out = g("embed.out.0.weight").view(D, D, F_out).permute(0, 2, 1).reshape(D, F_out * D) * math.sqrt(D)
        self.sub_out = _Lin(w_out, g("embed.out.0.bias") * math.sqrt(D), dev)
        self.layers = []
        for i in range(self.nb):
            p = f"encoders.{i}."
            wq, bq = g(p + "self_attn.linear_q.weight"), g(p + "self_attn.linear_q.bias")
            u, v = g(p + "self_attn.pos_bias_u"), g(p + "self_attn.pos_bias_v")
            # one projection GEMM: per head [q + u (dk) | q + v (dk)], then k (D), then v (D)
            wqq = torch.stack([wq.view(H, dk, D), wq.view(H, dk, D)], 1).reshape(2 * D, D)
            bqq = torch.stack([bq.view(H, dk) + u, bq.view(H, dk) + v], 1).reshape(2 * D)
            w_all = torch.cat([wqq, g(p + "self_attn.linear_k.weight"), g(p + "self_attn.linear_v.weight")], 0)
            b_all = torch.cat([bqq, g(p + "self_attn.linear_k.bias"), g(p + "self_attn.linear_v.bias")], 0)
            L = dict(
                qkv=_Lin(w_all, b_all, dev), pos=_Lin(g(p + "self_attn.linear_pos.weight"), None, dev),
                out=_Lin(g(p + "self_attn.linear_out.weight"), g(p + "self_attn.linear_out.bias"), dev),
                pw1=_Lin(g(p + "conv_module.pointwise_conv1.weight").squeeze(-1), g(p + "conv_module.pointwise_conv1.bias"), dev),
                dw_w=g(p + "conv_module.depthwise_conv.weight").squeeze(1).to(dev).contiguous(), dw_b=dv(p + "conv_module.depthwise_conv.bias"),
                pw2=_Lin(g(p + "conv_module.pointwise_conv2.weight").squeeze(-1), g(p + "conv_module.pointwise_conv2.bias"), dev),
                w1=_Lin(g(p + "feed_forward.w_1.weight"), g(p + "feed_forward.w_1.bias"), dev),
                w2=_Lin(g(p + "feed_forward.w_2.weight"), g(p + "feed_forward.w_2.bias"), dev))
            for n in ("norm_ff", "norm_mha", "norm_conv", "norm_final", "conv_module.norm"):
                L[n] = (dv(p + n + ".weight"), dv(p + n + ".bias"))
            self.layers.append(L)
        self.after_norm = (dv("after_norm.weight"), dv("after_norm.bias"))
        self._loaded = True
        return self

    def _lin(self, x, lin: _Lin):
        if x.shape[1] != lin.k:
            x = torch.nn.functional.pad(x, (0, lin.k - x.shape[1]))
        return self.ops.linear(x.contiguous(), lin.wp, lin.b, lin.n_out)

    def forward_packed(self, xs: torch.Tensor, xs_lens: Sequence[int]):
        """xs (B, T, input_size) f32, xs_lens -> packed rows [sum T'_b][D] of the valid output frames, T'_b = (len_b - 1) // 2."""
        if not self._loaded:
            raise RuntimeError("ConformerEncoder: load_state_dict() first")
        dev, D, H, ops = self.device, self.D, self.H, self.ops
        dk = D // H
        xs = xs.to(dev, torch.float32)
        lens = [int(v) for v in xs_lens]
        if xs.dim() != 3 or xs.shape[2] != self.idim or len(lens) != xs.shape[0] or max(lens) > xs.shape[1] or min(lens) < 0:
            raise ValueError(f"ConformerEncoder: expected xs (B, T, {self.idim}) with lengths <= T, got {tuple(xs.shape)} and {lens}")
        t_out = [max(0, (n - 1) // 2) for n in lens]
        (tok_seq, tok_t, start, Tt), n = _tables(t_out, dev)
        if n == 0:
            return torch.zeros(0, D, device=dev), t_out, (tok_seq, tok_t, start, Tt)
        F_out = (self.idim - 1) // 2
        # gather the 3 x 3 patches of every valid output frame: rows (frame, f'), columns (i, j)   [data movement only]
        b_idx, t_idx = tok_seq.long(), tok_t.long()
        fi = torch.arange(F_out, device=dev)
        rows = xs[b_idx[:, None], (2 * t_idx)[:, None] + torch.arange(3, device=dev)[None, :]]                              # (n, 3, idim)
        patches = torch.stack([rows[:, :, 2 * fi + j] for j in range(3)], dim=-1)                                          # (n, 3, F', 3)
        patches = patches.permute(0, 2, 1, 3).reshape(n * F_out, 9)
        h = ops.act_(self._lin(patches, self.sub_conv), 0)                           # (n * F', D), ReLU
        x = self._lin(h.view(n, F_out * D), self.sub_out)                            # (n, D), already x sqrt(D)
        t_max = max(t_out)
        pos_emb = _pos_table(D, t_max).to(dev)
        kstart, klen = start[b_idx].contiguous(), Tt[b_idx].contiguous()
        scale = 1.0 / math.sqrt(dk)
        for L in self.layers:
            hh = layernorm(x, *L["norm_mha"])
            qkv = self._lin(hh, L["qkv"])                                            # [n][2D | D | D]
            p = self._lin(pos_emb, L["pos"])                                         # [t_max][D]
            q2 = qkv[:, : 2 * D].contiguous()                                        # per head [q + u | q + v]
            k2 = torch.cat([qkv[:, 2 * D: 3 * D].view(n, H, dk), p[t_idx].view(n, H, dk)], dim=2).contiguous()   # per head [k | p]
            v = qkv[:, 3 * D:].contiguous()
            att = ops.attention(q2, k2, v, kstart, klen, H, 2 * dk, dk, scale)
            x = ops.add_(x, self._lin(att, L["out"]))
            hh = layernorm(x, *L["norm_conv"])
            hh = ops.glu(self._lin(hh, L["pw1"]), 0)
            hh = ops.dwconv(hh, L["dw_w"], L["dw_b"], tok_seq, tok_t, Tt, self.k)
            hh = ops.act_(layernorm(hh, *L["conv_module.norm"]), 1)
            x = ops.add_(x, self._lin(hh, L["pw2"]))
            hh = ops.act_(self._lin(layernorm(x, *L["norm_ff"]), L["w1"]), 1)
            x = ops.add_(x, self._lin(hh, L["w2"]))
            x = layernorm(x, *L["norm_final"])
        return layernorm(x, *self.after_norm), t_out, (tok_seq, tok_t, start, Tt)

    def forward(self, xs: torch.Tensor, xs_lens: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """The reference's signature: (B, T', D) padded output (rows beyond a prompt's own frames are zero: the reference leaves
        values there that every consumer masks) and the (B, 1, T') bool mask `masks[:, :, 2::2]` (conformer_encoder.py:398-433)."""
        x, t_out, (tok_seq, tok_t, _, _) = self.forward_packed(xs, xs_lens)
        B, T = xs.shape[0], xs.shape[1]
        t_pad = (T - 1) // 2
        out = torch.zeros(B, t_pad, self.D, dtype=torch.float32, device=self.device)
        out[tok_seq.long(), tok_t.long()] = x
        mask = torch.arange(t_pad, device=self.device)[None, :] < torch.tensor(t_out, device=self.device)[:, None]
        return out, mask.unsqueeze(1)

    __call__ = forward


class PerceiverResampler:
    """indextts/gpt/perceiver.py PerceiverResampler (depth 2, cross attention over [latents | context], GEGLU feed-forward, RMSNorm)."""

    def __init__(self, dim, depth=2, dim_context=None, num_latents=32, dim_head=64, heads=8, ff_mult=4, use_flash_attn=False, device="cuda:0"):
        self.dim, self.depth, self.dim_context = dim, depth, dim if dim_context is None else dim_context
        self.num_latents, self.dim_head, self.heads, self.ff_mult = num_latents, dim_head, heads, ff_mult
        if dim % 16 or dim_head % 4:
            raise ValueError("PerceiverResampler (HIP engine): dim % 16 == 0 and dim_head % 4 == 0 required")
        self.device = torch.device(device)
        self.ops = _Ops(self.device)
        self._loaded = False

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        dev = self.device
        g = lambda k: sd[k].detach().float().cpu()
        self.latents = g("latents").to(dev).contiguous()
        self.gamma = g("norm.gamma").to(dev).contiguous()
        self.proj = _Lin(g("proj_context.weight"), g("proj_context.bias"), dev) if "proj_context.weight" in sd else None
        self.layers = []
        for i in range(self.depth):
            p = f"layers.{i}."
            self.layers.append(dict(q=_Lin(g(p + "0.to_q.weight"), None, dev), kv=_Lin(g(p + "0.to_kv.weight"), None, dev),
                                    out=_Lin(g(p + "0.to_out.weight"), None, dev), ff1=_Lin(g(p + "1.0.weight"), g(p + "1.0.bias"), dev),
                                    ff2=_Lin(g(p + "1.2.weight"), g(p + "1.2.bias"), dev)))
        self._loaded = True
        return self

    def _lin(self, x, lin: _Lin):
        if x.shape[1] != lin.k:
            x = torch.nn.functional.pad(x, (0, lin.k - x.shape[1]))
        return self.ops.linear(x.contiguous(), lin.wp, lin.b, lin.n_out)

    def forward_packed(self, ctx: torch.Tensor, ctx_lens: Sequence[int]) -> torch.Tensor:
        """ctx: packed context rows [sum len_b][dim_context] -> (B, num_latents, dim)"""
        if not self._loaded:
            raise RuntimeError("PerceiverResampler: load_state_dict() first")
        dev, ops, nl, H, dh = self.device, self.ops, self.num_latents, self.heads, self.dim_head
        B = len(ctx_lens)
        x = self._lin(ctx, self.proj) if self.proj is not None else ctx
        lat = self.latents.unsqueeze(0).expand(B, -1, -1).reshape(B * nl, self.dim).contiguous()
        lens = torch.as_tensor([int(v) for v in ctx_lens], dtype=torch.int64)
        # key rows of batch b: its latents followed by its context rows   (cross_attn_include_queries)
        kv_len = lens + nl
        kv_start = torch.cumsum(kv_len, 0) - kv_len
        ctx_start = torch.cumsum(lens, 0) - lens
        lat_dst = (kv_start[:, None] + torch.arange(nl)[None, :]).reshape(-1).to(dev)
        ctx_dst = torch.cat([kv_start[b] + nl + torch.arange(int(lens[b])) for b in range(B)]).to(dev) if int(lens.sum()) else torch.zeros(0, dtype=torch.int64, device=dev)
        n_kv = int(kv_len.sum())
        kstart = kv_start.repeat_interleave(nl).to(dev, torch.int32).contiguous()
        klen = kv_len.repeat_interleave(nl).to(dev, torch.int32).contiguous()
        inner = H * dh
        for L in self.layers:
            rows = torch.empty(n_kv, self.dim, dtype=torch.float32, device=dev)
            rows[lat_dst] = lat
            if ctx_dst.numel():
                rows[ctx_dst] = x
            q = self._lin(lat, L["q"])
            kv = self._lin(rows, L["kv"])
            att = ops.attention(q, kv[:, :inner].contiguous(), kv[:, inner:].contiguous(), kstart, klen, H, dh, dh, dh ** -0.5)
            lat = ops.add_(self._lin(att, L["out"]), lat)
            lat = ops.add_(self._lin(ops.glu(self._lin(lat, L["ff1"]), 1), L["ff2"]), lat)
        return ops.l2norm_(lat, self.gamma, self.dim ** 0.5).view(B, nl, self.dim)

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The reference's signature: x (B, T, dim_context), mask (B, num_latents + T) bool with a True prefix per row."""
        B, T = x.shape[0], x.shape[1]
        lens = [T] * B if mask is None else [int(v) - self.num_latents for v in mask.sum(dim=1).tolist()]
        packed = torch.cat([x[b, : lens[b]] for b in range(B)], 0).to(self.device, torch.float32).contiguous()
        return self.forward_packed(packed, lens)

    __call__ = forward


class ConditioningEncoders:
    """`get_conditioning` / `get_emo_conditioning` / `get_emovec` / `merge_emovec` of UnifiedVoice (model_v2.py:556-593,827-838) on the
    engine; built from the `condition_module` / `emo_condition_module` sections of the reference config and the matching slices of
    the GPT checkpoint (`conditioning_encoder.*`, `perceiver_encoder.*`, `emo_conditioning_encoder.*`, `emo_perceiver_encoder.*`,
    `emovec_layer.*`, `emo_layer.*`)."""

    def __init__(self, model_dim: int, condition_module: Optional[dict], emo_condition_module: dict, cond_num: int = 32, device="cuda:0"):
        self.device = torch.device(device)
        self.model_dim = model_dim

        def pair(cm, dim, n_lat):
            enc = ConformerEncoder(input_size=1024 if "input_size" not in cm else cm["input_size"], output_size=cm["output_size"],
                                   linear_units=cm["linear_units"], attention_heads=cm["attention_heads"], num_blocks=cm["num_blocks"],
                                   input_layer=cm["input_layer"], device=device)
            per = PerceiverResampler(dim, dim_context=cm["output_size"], ff_mult=cm["perceiver_mult"], heads=cm["attention_heads"],
                                     num_latents=n_lat, device=device)
            return enc, per
        self.spk = pair(condition_module, model_dim, cond_num) if condition_module is not None else None
        self.emo = pair(emo_condition_module, emo_condition_module.get("perceiver_dim", 1024), 1)
        self.ops = _Ops(self.device)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        if self.spk is not None:
            self.spk[0].load_state_dict(sub("conditioning_encoder."))
            self.spk[1].load_state_dict(sub("perceiver_encoder."))
        self.emo[0].load_state_dict(sub("emo_conditioning_encoder."))
        self.emo[1].load_state_dict(sub("emo_perceiver_encoder."))
        self.emovec_layer = _Lin(sd["emovec_layer.weight"], sd["emovec_layer.bias"], self.device)
        self.emo_layer = _Lin(sd["emo_layer.weight"], sd["emo_layer.bias"], self.device)
        return self

    @staticmethod
    def _run(pair, feats, lens):
        # the reference passes the feature WIDTH (1024) as the "length" (infer_v2_5.py:760-765, infer_v2.py:643-648, model_v2.py:761;
        # SURVEY.md section 9 item 9) and its padding mask compares frame index < length, so a length above T simply means "every frame
        # is valid": clamp instead of rejecting it
        T = int(feats.shape[1])
        x, t_out, _ = pair[0].forward_packed(feats, [min(int(v), T) for v in lens])
        return pair[1].forward_packed(x, t_out)

    def get_conditioning(self, speech_conditioning_input: torch.Tensor, cond_mel_lengths) -> torch.Tensor:
        """(B, 1024, T) features (the reference passes them channel-first and transposes, model_v2.py:563) -> (B, 32, model_dim)"""
        return self._run(self.spk, speech_conditioning_input.transpose(1, 2), cond_mel_lengths)

    def get_emo_conditioning(self, speech_conditioning_input: torch.Tensor, cond_mel_lengths) -> torch.Tensor:
        return self._run(self.emo, speech_conditioning_input.transpose(1, 2), cond_mel_lengths).squeeze(1)

    def get_emovec(self, emo_speech_conditioning_latent: torch.Tensor, emo_cond_lengths) -> torch.Tensor:
        """model_v2.py:827-831: (B, T, 1024) features -> (B, model_dim)"""
        v = self.get_emo_conditioning(emo_speech_conditioning_latent.transpose(1, 2), emo_cond_lengths)
        lin = lambda x, l: self.ops.linear(x.contiguous(), l.wp, l.b, l.n_out)
        return lin(lin(v, self.emovec_layer), self.emo_layer)

    def merge_emovec(self, speech_conditioning_latent, emo_speech_conditioning_latent, cond_lengths, emo_cond_lengths, alpha=1.0):
        emo_vec = self.get_emovec(emo_speech_conditioning_latent, emo_cond_lengths)          # model_v2.py:833-838
        base_vec = self.get_emovec(speech_conditioning_latent, cond_lengths)
        return base_vec + alpha * (emo_vec - base_vec)
