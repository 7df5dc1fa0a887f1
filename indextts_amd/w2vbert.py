"""w2v-bert-2.0 feature encoder on the HIP engine: the `semantic_model` behind `IndexTTS2.get_emb`
(indextts/infer_v2_5.py:171-176 builds `Wav2Vec2BertModel.from_pretrained("facebook/w2v-bert-2.0")`, :282-290 take
`hidden_states[17]` and normalise it with the wav2vec2bert statistics).  The model class lives in the `transformers` dependency
(modeling_wav2vec2_bert.py); this is a host mirror of its inference path with the same parameter names:

    feature_projection (LayerNorm -> Linear)  ->  N x Wav2Vec2BertEncoderLayer:
        x += 0.5 * FFN1(LN(x));  x += SelfAttention(LN(x)) [relative_key];  x += ConvModule(x);  x += 0.5 * FFN2(LN(x));  x = LN(x)

Once-per-prompt work on exact-f32 unit ops of the C ABI over PACKED rows (only the valid frames of every prompt; the class's key
masks and zeroed padding have nothing to act on, and because the depthwise conv is causal a prompt's valid frames do not depend
on the batch it sits in -- tools/make_golden_w2vbert.py prints that check): dense layers and the 1x1 convs on `itts_gemm_forward`
(q, k, v as one GEMM), LayerNorms on `itts_layernorm_forward`, the relative-key attention on `itts_attention_relkey_forward`
(q . dist_emb[r] for the 73 clamped distances once per query, then added to q . k_j), GLU / swish on `itts_tok_{glu,act}_forward`,
the causal depthwise conv on `itts_tok_dwconv_causal_forward`, residuals on `itts_tok_scale_residual_forward`.  Only the layers up
to the tapped hidden state run (17 of 24 in the pipeline).
"""
import math
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from .codec import _tables
from .cond import _Lin, _Ops
from .gpt import layernorm


class _WOps(_Ops):
    def attention_relkey(self, q, k, v, kstart, klen, qpos, dist_emb, left, right, heads, dq, dv, scale):
        out = torch.empty(q.shape[0], heads * dv, dtype=torch.float32, device=self.device)
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_attention_relkey_forward(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(out), _lib.ptr(kstart), _lib.ptr(klen),
                                                            _lib.ptr(qpos), _lib.ptr(dist_emb), left, right, q.shape[0], heads, dq, dv, float(scale),
                                                            self._st()), "itts_attention_relkey_forward")
        return out

    def dwconv_causal(self, x, w, tok_seq, tok_t, seq_T, k):
        y = torch.empty_like(x)
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_dwconv_causal_forward(_lib.ptr(x), _lib.ptr(w), None, _lib.ptr(y), _lib.ptr(tok_seq), _lib.ptr(tok_t),
                                                             _lib.ptr(seq_T), x.shape[0], x.shape[1], k, self._st()), "itts_tok_dwconv_causal_forward")
        return y


class Wav2Vec2BertModel:
    """Constructor arguments follow `transformers.Wav2Vec2BertConfig`; only the w2v-bert-2.0 architecture is built (relative_key
    positions, swish, causal depthwise conv, no adapter, eval mode)."""

    def __init__(self, hidden_size: int = 1024, num_hidden_layers: int = 24, num_attention_heads: int = 16, intermediate_size: int = 4096,
                 feature_projection_input_dim: int = 160, position_embeddings_type: str = "relative_key", left_max_position_embeddings: int = 64,
                 right_max_position_embeddings: int = 8, conv_depthwise_kernel_size: int = 31, hidden_act: str = "swish",
                 layer_norm_eps: float = 1e-5, add_adapter: bool = False, device="cuda:0", **_unused):
        if position_embeddings_type != "relative_key" or hidden_act not in ("swish", "silu") or add_adapter:
            raise NotImplementedError("Wav2Vec2BertModel (HIP engine): relative_key / swish / no adapter only")
        dh = hidden_size // num_attention_heads
        if hidden_size % 64 or dh * num_attention_heads != hidden_size or dh % 4 or dh > 128 or intermediate_size % 16:
            raise ValueError("Wav2Vec2BertModel (HIP engine): hidden_size % 64 == 0, head_dim % 4 == 0 and <= 128, intermediate_size % 16 == 0 required")
        if left_max_position_embeddings + right_max_position_embeddings + 1 > 512:
            raise ValueError("Wav2Vec2BertModel (HIP engine): at most 512 relative distances")
        self.D, self.nl, self.H, self.I, self.idim = hidden_size, num_hidden_layers, num_attention_heads, intermediate_size, feature_projection_input_dim
        self.left, self.right, self.k, self.eps = left_max_position_embeddings, right_max_position_embeddings, conv_depthwise_kernel_size, layer_norm_eps
        self.device = torch.device(device)
        self.ops = _WOps(self.device)
        self._loaded = False

    def eval(self):
        return self

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True, n_layers: Optional[int] = None):
        """`n_layers` keeps only the first layers resident (the pipeline taps hidden_states[17])."""
        dev = self.device
        g = lambda k: sd[k].detach().float().cpu()
        dv = lambda k: g(k).to(dev).contiguous()
        ln = lambda p: (dv(p + ".weight"), dv(p + ".bias"))
        self.fp_norm = ln("feature_projection.layer_norm")
        self.fp = _Lin(g("feature_projection.projection.weight"), g("feature_projection.projection.bias"), dev)
        self.layers: List[dict] = []
        for i in range(self.nl if n_layers is None else min(self.nl, n_layers)):
            p = f"encoder.layers.{i}."
            a = p + "self_attn."
            L = dict(
                qkv=_Lin(torch.cat([g(a + f"linear_{n}.weight") for n in "qkv"], 0), torch.cat([g(a + f"linear_{n}.bias") for n in "qkv"], 0), dev),
                out=_Lin(g(a + "linear_out.weight"), g(a + "linear_out.bias"), dev), dist=dv(a + "distance_embedding.weight"),
                pw1=_Lin(g(p + "conv_module.pointwise_conv1.weight").squeeze(-1), None, dev),
                dw_w=g(p + "conv_module.depthwise_conv.weight").squeeze(1).to(dev).contiguous(),
                pw2=_Lin(g(p + "conv_module.pointwise_conv2.weight").squeeze(-1), None, dev))
            for ff in ("ffn1", "ffn2"):
                L[ff + "_in"] = _Lin(g(p + ff + ".intermediate_dense.weight"), g(p + ff + ".intermediate_dense.bias"), dev)
                L[ff + "_out"] = _Lin(g(p + ff + ".output_dense.weight"), g(p + ff + ".output_dense.bias"), dev)
            for n in ("ffn1_layer_norm", "self_attn_layer_norm", "conv_module.layer_norm", "conv_module.depthwise_layer_norm", "ffn2_layer_norm",
                      "final_layer_norm"):
                L[n] = ln(p + n)
            if L["dist"].shape != (self.left + self.right + 1, self.D // self.H):
                raise ValueError(f"distance_embedding of layer {i}: {tuple(L['dist'].shape)}")
            self.layers.append(L)
        self.half = torch.full((self.D,), 0.5, dtype=torch.float32, device=dev)
        self._loaded = True
        return self

    def _lin(self, x, lin: _Lin):
        if x.shape[1] != lin.k:
            x = torch.nn.functional.pad(x, (0, lin.k - x.shape[1]))
        return self.ops.linear(x.contiguous(), lin.wp, lin.b, lin.n_out)

    def hidden_states_packed(self, input_features: torch.Tensor, lens: Sequence[int], n_layers: Optional[int] = None, keep: Optional[Sequence[int]] = None):
        """input_features (B, T, idim), lens -> (dict layer index -> packed rows [sum len_b][D], tables).  `keep` = the hidden-state
        indices to return (0 = after the feature projection); default only the last computed one."""
        if not self._loaded:
            raise RuntimeError("Wav2Vec2BertModel: load_state_dict() first")
        dev, D, H, ops, eps = self.device, self.D, self.H, self.ops, self.eps
        dh = D // H
        n_layers = len(self.layers) if n_layers is None else int(n_layers)
        if n_layers > len(self.layers):
            raise ValueError(f"Wav2Vec2BertModel: hidden state {n_layers} asked, {len(self.layers)} layers loaded")
        x_in = input_features.to(dev, torch.float32)
        lens = [int(v) for v in lens]
        if x_in.dim() != 3 or x_in.shape[2] != self.idim or len(lens) != x_in.shape[0] or max(lens) > x_in.shape[1] or min(lens) < 0:
            raise ValueError(f"Wav2Vec2BertModel: expected input_features (B, T, {self.idim}) with lengths <= T, got {tuple(x_in.shape)} and {lens}")
        keep = {n_layers} if keep is None else set(int(v) for v in keep)
        (tok_seq, tok_t, start, Tt), n = _tables(lens, dev)
        tabs = (tok_seq, tok_t, start, Tt)
        if n == 0:
            return {i: torch.zeros(0, D, device=dev) for i in keep}, tabs
        b_idx = tok_seq.long()
        x = self._lin(layernorm(x_in[b_idx, tok_t.long()].contiguous(), *self.fp_norm, eps=eps), self.fp)
        outs = {0: x.clone()} if 0 in keep else {}
        kstart, klen = start[b_idx].contiguous(), Tt[b_idx].contiguous()
        scale = 1.0 / math.sqrt(dh)
        for li, L in enumerate(self.layers[:n_layers]):
            h = ops.act_(self._lin(layernorm(x, *L["ffn1_layer_norm"], eps=eps), L["ffn1_in"]), 1)
            x = ops.scale_residual_(x, self._lin(h, L["ffn1_out"]), self.half)
            qkv = self._lin(layernorm(x, *L["self_attn_layer_norm"], eps=eps), L["qkv"])
            att = ops.attention_relkey(qkv[:, :D].contiguous(), qkv[:, D: 2 * D].contiguous(), qkv[:, 2 * D:].contiguous(), kstart, klen, tok_t,
                                       L["dist"], self.left, self.right, H, dh, dh, scale)
            x = ops.add_(x, self._lin(att, L["out"]))
            h = ops.glu(self._lin(layernorm(x, *L["conv_module.layer_norm"], eps=eps), L["pw1"]), 0)
            h = ops.dwconv_causal(h, L["dw_w"], tok_seq, tok_t, Tt, self.k)
            h = ops.act_(layernorm(h, *L["conv_module.depthwise_layer_norm"], eps=eps), 1)
            x = ops.add_(x, self._lin(h, L["pw2"]))
            h = ops.act_(self._lin(layernorm(x, *L["ffn2_layer_norm"], eps=eps), L["ffn2_in"]), 1)
            x = ops.scale_residual_(x, self._lin(h, L["ffn2_out"]), self.half)
            x = layernorm(x, *L["final_layer_norm"], eps=eps)
            if li + 1 in keep:
                outs[li + 1] = x.clone() if li + 1 < n_layers else x
        return outs, tabs

    @staticmethod
    def _lens(input_features, attention_mask):
        if attention_mask is None:
            return [input_features.shape[1]] * input_features.shape[0]
        m = attention_mask.to(torch.int64)
        lens = m.sum(dim=1)
        if not bool((m == (torch.arange(m.shape[1], device=m.device)[None, :] < lens[:, None]).to(torch.int64)).all()):
            raise ValueError("Wav2Vec2BertModel (HIP engine): attention_mask rows must be a run of ones followed by zeros (right padding)")
        return lens.tolist()

    def _pad(self, rows, tabs, B, T):
        out = torch.zeros(B, T, self.D, dtype=torch.float32, device=self.device)
        out[tabs[0].long(), tabs[1].long()] = rows
        return out

    def forward(self, input_features: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, output_hidden_states: bool = False, **_unused):
        """The class's call signature: `.last_hidden_state` and, with output_hidden_states, `.hidden_states` (a tuple of (B, T, D);
        rows of padded frames are zero -- the class leaves masked-out values there)."""
        B, T = input_features.shape[0], input_features.shape[1]
        n = len(self.layers)
        outs, tabs = self.hidden_states_packed(input_features, self._lens(input_features, attention_mask), n,
                                               keep=range(n + 1) if output_hidden_states else [n])
        hs = tuple(self._pad(outs[i], tabs, B, T) for i in sorted(outs))
        return SimpleNamespace(last_hidden_state=hs[-1], hidden_states=hs if output_hidden_states else None)

    __call__ = forward

    def get_emb(self, input_features, attention_mask, semantic_mean, semantic_std, layer: int = 17):
        """`IndexTTS2.get_emb` (infer_v2_5.py:282-290): (hidden_states[layer] - mean) / std, (B, T, D)"""
        B, T = input_features.shape[0], input_features.shape[1]
        outs, tabs = self.hidden_states_packed(input_features, self._lens(input_features, attention_mask), layer)
        feat = self._pad(outs[layer], tabs, B, T)
        return (feat - semantic_mean.to(self.device)) / semantic_std.to(self.device)
