"""Host-side mirrors of the two modules between the GPT codes and the flow-matching decoder, backed by the HIP engine:

  * `EnhancedCodec.decode(codes)` (indextts/codec/models.py:205-231): codebook lookup + weight-normed projection
    (FVQ.vq2emb), Vocos ConvNeXt backbone (indextts/codec/kmeans/vocos.py:468-526,719-782), nearest x2 upsampling + `up`
    conv; call site indextts/infer_v2_5.py:832.
  * `InterpolateRegulator.forward(x, ylens=...)` (indextts/s2mel/modules/length_regulator.py:90-141, continuous input,
    no f0 / VQ): content_in_proj, nearest interpolation to the mel length, 4 x (Conv1d k=3, GroupNorm(1), Mish), 1x1 conv;
    call sites indextts/infer_v2_5.py:651-656,835-838.

The classes sequence C-ABI calls (`itts_vq_project_forward`, `itts_tok_*_forward`, `itts_gemm_forward`,
`itts_layernorm_forward`); all arithmetic is f32 (the stage is ~40 GFLOP per utterance).  Utterances of a batch are packed
back to back, each with its own length (the reference runs the stage at batch 1): padding never reaches a conv or a norm.
"""
import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from . import _lib
from .bigvgan import fold_weight_norm
from .gpt import gemm as engine_gemm
from .gpt import layernorm as engine_layernorm
from .gpt import pack_gemm_weight

F32 = 0


def _tables(lens: Sequence[int], device):
    """(tok_seq, tok_t, start, T) int32 device tensors of packed sequences of the given lengths, and the row count.  Built on the device from the
    lengths (the row tables are not computed on the host and copied: the device idles meanwhile, profiles/r06f/gap_report)."""
    Th = torch.as_tensor(list(lens), dtype=torch.int32)
    n = int(Th.sum())
    T = Th.to(device)
    start = torch.cumsum(T, 0, dtype=torch.int32) - T
    tok_seq = torch.repeat_interleave(torch.arange(T.numel(), dtype=torch.int32, device=device), T.long(), output_size=n)
    tok_t = torch.arange(n, dtype=torch.int32, device=device) - start[tok_seq.long()]
    return tuple(t.contiguous() for t in (tok_seq, tok_t, start, T)), n


def _conv_matrix(w: torch.Tensor) -> torch.Tensor:
    """Conv1d weight [C_out][C_in][k] -> [k*C_in][C_out] (row index j*C_in + c: the im2col column order)"""
    co, ci, k = w.shape
    return w.permute(2, 1, 0).reshape(k * ci, co).contiguous()


class _TokOps:
    """thin wrappers over the C ABI, all on packed f32 [n][C] matrices"""

    def __init__(self, device):
        self.device = torch.device(device)
        self.L = _lib.lib()

    def _st(self):
        return _lib.stream_ptr(self.device)

    def linear(self, x, wp, bias, n_out):
        return engine_gemm(x, wp, bias, n_out, F32, prefill_tiles=True)

    def conv(self, x, tabs_dst, n_dst, src_start, src_T, dst_T, k, wp, bias, c_out):
        """`same` Conv1d over the source sequences nearest-interpolated to the destination lengths"""
        tok_seq, tok_t = tabs_dst
        Cc = x.shape[1]
        col = torch.empty(n_dst, k * Cc, dtype=torch.float32, device=self.device)
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_gather_conv_forward(_lib.ptr(x), _lib.ptr(col), _lib.ptr(tok_seq), _lib.ptr(tok_t),
                                                           _lib.ptr(src_start), _lib.ptr(src_T), _lib.ptr(dst_T), n_dst, Cc, k,
                                                           self._st()), "itts_tok_gather_conv_forward")
        return self.linear(col, wp, bias, c_out)

    def dwconv(self, x, w, b, tok_seq, tok_t, seq_T, k):
        y = torch.empty_like(x)
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_dwconv_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), _lib.ptr(tok_seq),
                                                      _lib.ptr(tok_t), _lib.ptr(seq_T), x.shape[0], x.shape[1], k, self._st()),
                       "itts_tok_dwconv_forward")
        return y

    def gelu_(self, x):
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_gelu_forward(_lib.ptr(x), x.numel(), self._st()), "itts_tok_gelu_forward")
        return x

    def scale_residual_(self, x, y, gamma):
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_scale_residual_forward(_lib.ptr(x), _lib.ptr(y), _lib.ptr(gamma), x.shape[0], x.shape[1],
                                                              self._st()), "itts_tok_scale_residual_forward")
        return x

    def groupnorm_mish_(self, x, gamma, beta, start, T, eps=1e-5):
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_groupnorm_mish_forward(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(start),
                                                              _lib.ptr(T), int(T.numel()), x.shape[1], float(eps), self._st()),
                       "itts_tok_groupnorm_mish_forward")
        return x


class EnhancedCodec:
    """The reference's semantic codec on the engine: `decode` / `vq2emb` (codes -> features) and, when the checkpoint carries the
    encoder half, `quantize` (features -> codes + quantized features; the v2 pipeline's prompt side)."""

    def __init__(self, codebook_size=8192, hidden_size=1024, codebook_dim=8, vocos_dim=384, vocos_intermediate_dim=2048,
                 vocos_num_layers=12, device="cuda:0", **_unused):
        self.codebook_size, self.hidden_size, self.codebook_dim = codebook_size, hidden_size, codebook_dim
        self.vocos_dim, self.vocos_intermediate_dim, self.vocos_num_layers = vocos_dim, vocos_intermediate_dim, vocos_num_layers
        if hidden_size % 16 or vocos_dim % 64 or vocos_intermediate_dim % 16:
            raise ValueError("EnhancedCodec (HIP engine): hidden_size % 16, vocos_dim % 64, vocos_intermediate_dim % 16 must be 0")
        self.device = torch.device(device)
        self._p: Dict[str, torch.Tensor] = {}
        self._loaded = False

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        dev = self.device
        sd = fold_weight_norm({k: v for k, v in sd.items()})
        f = lambda t: t.detach().to(dev, torch.float32).contiguous()
        pk = lambda kn: pack_gemm_weight(kn.detach().float().cpu(), F32).to(dev)
        p = self._p
        Q = "quantizer.quantizers.0."
        p["codebook"] = f(sd[Q + "codebook.weight"])
        w = sd[Q + "out_project.weight"]
        p["out_w"], p["out_b"] = f(w.reshape(w.shape[0], -1)), f(sd[Q + "out_project.bias"])
        self._load_backbone(sd, "decoder.", "", f, pk)
        p["up_w"], p["up_b"] = pk(_conv_matrix(sd["up.weight"])), f(sd["up.bias"])
        # the quantize half (stride-2 down conv, Vocos encoder, FVQ in_project + search), when the checkpoint carries it
        self._has_encoder = all(k in sd for k in ("down.weight", "encoder.0.embed.weight", Q + "in_project.weight"))
        if self._has_encoder:
            if self.codebook_dim > 16:
                raise ValueError("EnhancedCodec.quantize (HIP engine): codebook_dim <= 16")
            p["down_w"], p["down_b"] = pk(_conv_matrix(sd["down.weight"])), f(sd["down.bias"])
            self._load_backbone(sd, "encoder.", "enc.", f, pk)
            wi = sd[Q + "in_project.weight"]
            p["in_w"], p["in_b"] = f(wi.reshape(wi.shape[0], -1)), f(sd[Q + "in_project.bias"])
            cbn = torch.nn.functional.normalize(sd[Q + "codebook.weight"].detach().float().cpu())        # F.normalize(codebook), decode_latents
            p["cb_norm"], p["cb_sq"] = f(cbn), f(cbn.pow(2).sum(1))
        self._loaded = True
        used = ("decoder.", "up.", Q) + (("encoder.", "down.") if self._has_encoder else ())
        return [k for k in sd if not k.startswith(used)]

    def _load_backbone(self, sd, ref_prefix: str, key: str, f, pk):
        """VocosBackbone + the Linear after it (`decoder` / `encoder` of the reference module) under parameter keys `key` + name"""
        p, b = self._p, ref_prefix + "0."
        p[key + "embed_w"], p[key + "embed_b"] = pk(_conv_matrix(sd[b + "embed.weight"])), f(sd[b + "embed.bias"])
        p[key + "norm_g"], p[key + "norm_b"] = f(sd[b + "norm.weight"]), f(sd[b + "norm.bias"])
        for i in range(self.vocos_num_layers):
            c = f"{b}convnext.{i}."
            dw = sd[c + "dwconv.weight"]
            p[f"{key}dw_w{i}"], p[f"{key}dw_b{i}"] = f(dw.reshape(dw.shape[0], -1)), f(sd[c + "dwconv.bias"])
            p[f"{key}ln_g{i}"], p[f"{key}ln_b{i}"] = f(sd[c + "norm.weight"]), f(sd[c + "norm.bias"])
            p[f"{key}pw1_w{i}"], p[f"{key}pw1_b{i}"] = pk(sd[c + "pwconv1.weight"].t()), f(sd[c + "pwconv1.bias"])
            p[f"{key}pw2_w{i}"], p[f"{key}pw2_b{i}"] = pk(sd[c + "pwconv2.weight"].t()), f(sd[c + "pwconv2.bias"])
            p[f"{key}gamma{i}"] = f(sd[c + "gamma"])
        p[key + "fln_g"], p[key + "fln_b"] = f(sd[b + "final_layer_norm.weight"]), f(sd[b + "final_layer_norm.bias"])
        p[key + "dec1_w"], p[key + "dec1_b"] = pk(sd[ref_prefix + "1.weight"].t()), f(sd[ref_prefix + "1.bias"])

    def _backbone(self, key: str, e: torch.Tensor, tabs, n: int) -> torch.Tensor:
        """e [n][hidden] packed rows -> [n][hidden]: embed conv (k 7), LayerNorm, ConvNeXt blocks, final LayerNorm, Linear (vocos.py:770-782)"""
        tok_seq, tok_t, start, Tt = tabs
        p, ops, D = self._p, _TokOps(self.device), self.vocos_dim
        x = ops.conv(e, (tok_seq, tok_t), n, start, Tt, Tt, 7, p[key + "embed_w"], p[key + "embed_b"], D)
        x = engine_layernorm(x, p[key + "norm_g"], p[key + "norm_b"], eps=1e-6)
        for i in range(self.vocos_num_layers):                                                             # ConvNeXtBlock
            y = ops.dwconv(x, p[f"{key}dw_w{i}"], p[f"{key}dw_b{i}"], tok_seq, tok_t, Tt, 7)
            y = engine_layernorm(y, p[f"{key}ln_g{i}"], p[f"{key}ln_b{i}"], eps=1e-6)
            y = ops.gelu_(ops.linear(y, p[f"{key}pw1_w{i}"], p[f"{key}pw1_b{i}"], self.vocos_intermediate_dim))
            y = ops.linear(y, p[f"{key}pw2_w{i}"], p[f"{key}pw2_b{i}"], D)
            ops.scale_residual_(x, y, p[f"{key}gamma{i}"])
        x = engine_layernorm(x, p[key + "fln_g"], p[key + "fln_b"], eps=1e-6)
        return ops.linear(x, p[key + "dec1_w"], p[key + "dec1_b"], self.hidden_size)

    def eval(self):
        return self

    def to(self, device):
        return self

    @torch.no_grad()
    def vq2emb(self, codes: torch.Tensor) -> torch.Tensor:
        """`quantizer.vq2emb(codes)` (residual_vq.py:144-152 -> FVQ.vq2emb): codes (B, T) or (1, B, T) or (B, 1, T) int ->
        (B, hidden, T) f32, the layout the reference returns (indextts/infer_v2.py:657-658 transposes it)."""
        if not self._loaded:
            raise RuntimeError("EnhancedCodec: load_state_dict() first")
        if codes.dim() == 3:
            codes = codes[0] if codes.shape[0] == 1 else codes[:, 0]
        dev, p = self.device, self._p
        B, T = codes.shape
        flat = codes.reshape(-1).to(dev, torch.int64).contiguous()
        e = torch.empty(B * T, self.hidden_size, dtype=torch.float32, device=dev)
        if B * T:
            with _lib.on_device(dev):
                _lib.check(_lib.lib().itts_vq_project_forward(_lib.ptr(flat), _lib.ptr(p["codebook"]), _lib.ptr(p["out_w"]),
                                                              _lib.ptr(p["out_b"]), _lib.ptr(e), B * T, self.codebook_size,
                                                              self.codebook_dim, self.hidden_size, _lib.stream_ptr(dev)),
                           "itts_vq_project_forward")
        return e.reshape(B, T, self.hidden_size).transpose(1, 2)

    @torch.no_grad()
    def quantize(self, x: torch.Tensor, lens: Optional[Sequence[int]] = None):
        """`EnhancedCodec.quantize` (indextts/codec/models.py:179-199; `_, S_ref = semantic_codec.quantize(spk_cond_emb)`, indextts/infer_v2.py:465):
        x (B, T, hidden) features -> (indices, quantized (B, T', hidden)), T' = (T - 1) // 2 + 1; indices are (B, T') int64 -- for B == 1
        the reference's squeeze of the quantizer axis leaves the same (1, T').  With `lens` each row is encoded at its own length (rows of
        the outputs beyond (len - 1) // 2 + 1 are zero)."""
        if not self._loaded or not self._has_encoder:
            raise RuntimeError("EnhancedCodec.quantize: load a state dict that carries the encoder half (down.*, encoder.*, in_project) first")
        dev, p, ops, H, L = self.device, self._p, _TokOps(self.device), self.hidden_size, _lib.lib()
        B, T = x.shape[0], x.shape[1]
        lens = [T] * B if lens is None else [int(v) for v in lens]
        olens = [(v - 1) // 2 + 1 if v > 0 else 0 for v in lens]
        Tq = (T - 1) // 2 + 1 if T > 0 else 0
        idx = torch.zeros(B, Tq, dtype=torch.int64, device=dev)
        q = torch.zeros(B, Tq, H, dtype=torch.float32, device=dev)
        tabs, n = _tables(olens, dev)
        if n == 0:
            return idx, q
        # stride-2, k = 3, padding 1 conv as a row gather (data movement) + GEMM: output t reads input rows 2t - 1, 2t, 2t + 1 of its own sequence
        xd = x.to(dev, torch.float32)
        cols = []
        for b in range(B):
            if olens[b] == 0:
                continue
            xp = torch.nn.functional.pad(xd[b, : lens[b]], (0, 0, 1, 2))                  # one zero row in front, two behind (even lengths read one)
            t = torch.arange(olens[b], device=dev) * 2
            cols.append(torch.cat([xp[t], xp[t + 1], xp[t + 2]], dim=1))
        col = torch.cat(cols, 0).contiguous()
        h = ops.gelu_(ops.linear(col, p["down_w"], p["down_b"], H))
        h = self._backbone("enc.", h, tabs, n).contiguous()
        flat = torch.empty(n, dtype=torch.int64, device=dev)
        qf = torch.empty(n, H, dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            st = _lib.stream_ptr(dev)
            _lib.check(L.itts_vq_search_forward(_lib.ptr(h), _lib.ptr(p["in_w"]), _lib.ptr(p["in_b"]), _lib.ptr(p["cb_norm"]), _lib.ptr(p["cb_sq"]),
                                                _lib.ptr(flat), n, H, self.codebook_size, self.codebook_dim, st), "itts_vq_search_forward")
            _lib.check(L.itts_vq_project_forward(_lib.ptr(flat), _lib.ptr(p["codebook"]), _lib.ptr(p["out_w"]), _lib.ptr(p["out_b"]), _lib.ptr(qf),
                                                 n, self.codebook_size, self.codebook_dim, H, st), "itts_vq_project_forward")
        o = 0
        for b in range(B):
            idx[b, : olens[b]], q[b, : olens[b]] = flat[o:o + olens[b]], qf[o:o + olens[b]]
            o += olens[b]
        return idx, q

    @property
    def quantizer(self):
        return self                                                    # `semantic_codec.quantizer.vq2emb(...)` call sites

    @torch.no_grad()
    def decode(self, codes: torch.Tensor, code_lens: Optional[Sequence[int]] = None) -> torch.Tensor:
        """codes (B, T) or (1, B, T) int -> (B, 2T, hidden) f32; with `code_lens` each row is decoded at its own length (frames
        beyond 2 * len are zero)."""
        if not self._loaded:
            raise RuntimeError("EnhancedCodec: load_state_dict() first")
        if codes.dim() == 3:
            codes = codes[0]
        dev, p, ops = self.device, self._p, _TokOps(self.device)
        B, T = codes.shape
        lens = [T] * B if code_lens is None else [int(v) for v in code_lens]
        (tok_seq, tok_t, start, Tt), n = _tables(lens, dev)
        out = torch.zeros(B, 2 * T, self.hidden_size, dtype=torch.float32, device=dev)
        if n == 0:
            return out
        flat = torch.cat([codes[b, : lens[b]] for b in range(B)]).to(dev, torch.int64).contiguous()
        D, H, L = self.vocos_dim, self.hidden_size, _lib.lib()
        e = torch.empty(n, H, dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            _lib.check(L.itts_vq_project_forward(_lib.ptr(flat), _lib.ptr(p["codebook"]), _lib.ptr(p["out_w"]), _lib.ptr(p["out_b"]),
                                                 _lib.ptr(e), n, self.codebook_size, self.codebook_dim, H, _lib.stream_ptr(dev)),
                       "itts_vq_project_forward")
        x = self._backbone("", e, (tok_seq, tok_t, start, Tt), n)                                        # VocosBackbone + Linear
        (tseq2, tt2, start2, T2), n2 = _tables([2 * v for v in lens], dev)                                 # x2 nearest + up conv
        y = ops.conv(x, (tseq2, tt2), n2, start, Tt, T2, 3, p["up_w"], p["up_b"], H)
        o = 0
        for b in range(B):
            out[b, : 2 * lens[b]] = y[o:o + 2 * lens[b]]
            o += 2 * lens[b]
        return out


class InterpolateRegulator:
    def __init__(self, channels: int, sampling_ratios=(1, 1, 1, 1), is_discrete: bool = False, in_channels: Optional[int] = None,
                 vector_quantize: bool = False, codebook_size: int = 1024, out_channels: Optional[int] = None, groups: int = 1,
                 n_codebooks: int = 1, quantizer_dropout: float = 0.0, f0_condition: bool = False, n_f0_bins: int = 512,
                 device="cuda:0"):
        if is_discrete or vector_quantize or f0_condition or groups != 1 or (out_channels not in (None, channels)):
            raise NotImplementedError("engine implements the IndexTTS-2 regulator: continuous input, GroupNorm(1), no VQ / f0")
        if in_channels is None or in_channels % 16 or channels % 16:
            raise ValueError("InterpolateRegulator (HIP engine): in_channels and channels must be multiples of 16")
        self.channels, self.in_channels, self.n_layers = channels, in_channels, len(sampling_ratios)
        self.interpolate = self.n_layers > 0
        self.device = torch.device(device)
        self._p: Dict[str, torch.Tensor] = {}
        self._loaded = False

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        dev = self.device
        f = lambda t: t.detach().to(dev, torch.float32).contiguous()
        pk = lambda kn: pack_gemm_weight(kn.detach().float().cpu(), F32).to(dev)
        p = self._p
        p["in_w"], p["in_b"] = pk(sd["content_in_proj.weight"].t()), f(sd["content_in_proj.bias"])
        for i in range(self.n_layers):
            p[f"cw{i}"], p[f"cb{i}"] = pk(_conv_matrix(sd[f"model.{3 * i}.weight"])), f(sd[f"model.{3 * i}.bias"])
            p[f"g{i}"], p[f"b{i}"] = f(sd[f"model.{3 * i + 1}.weight"]), f(sd[f"model.{3 * i + 1}.bias"])
        p["ow"], p["ob"] = pk(_conv_matrix(sd[f"model.{3 * self.n_layers}.weight"])), f(sd[f"model.{3 * self.n_layers}.bias"])
        self._loaded = True
        return [k for k in sd if k in ("mask_token", "embedding.weight")]

    def eval(self):
        return self

    def to(self, device):
        return self

    @torch.no_grad()
    def forward(self, x: torch.Tensor, ylens: torch.Tensor = None, n_quantizers=None, f0=None, xlens: Optional[Sequence[int]] = None,
                frame_lens: Optional[Sequence[int]] = None):
        """x (B, T, in_channels), ylens (B,) -> (out (B, max ylens, channels) zero beyond each row's ylen, ylens, None, None, None).
        `xlens`: valid input frames per row (default T for every row).  `frame_lens`: frames each row is stretched to and
        processed at -- default max(ylens) for EVERY row, which is what the reference does with a batch (rows shorter than the
        longest are interpolated to the longest, normalised over it and cut by the mask afterwards, :117-141); the pipeline
        passes `frame_lens = ylens`, i.e. what a batch-1 reference call per utterance computes."""
        if not self._loaded:
            raise RuntimeError("InterpolateRegulator: load_state_dict() first")
        if f0 is not None:
            raise NotImplementedError("f0 conditioning is not part of the IndexTTS-2 regulator")
        dev, p, ops = self.device, self._p, _TokOps(self.device)
        B, T, _ = x.shape
        xl = [T] * B if xlens is None else [int(v) for v in xlens]
        yl = [int(v) for v in torch.as_tensor(ylens).reshape(-1)]
        if not self.interpolate:
            yl = [min(a, b) for a, b in zip(yl, xl)]
        Tm = max(yl) if yl else 0
        fl = ([Tm] * B if self.interpolate else list(xl)) if frame_lens is None else [int(v) for v in frame_lens]
        (_, _, sstart, sT), ns = _tables(xl, dev)
        (dseq, dt, dstart, dT), nd = _tables(fl, dev)
        Cc = self.channels
        out = torch.zeros(B, Tm, Cc, dtype=torch.float32, device=dev)
        if ns == 0 or nd == 0:
            return out, torch.as_tensor(yl, device=dev), None, None, None
        xs = torch.cat([x[b, : xl[b]] for b in range(B)], 0).to(dev, torch.float32).contiguous()
        h = ops.linear(xs, p["in_w"], p["in_b"], Cc)                                                       # content_in_proj
        src_start, src_T = sstart, sT
        for i in range(self.n_layers):
            h = ops.conv(h, (dseq, dt), nd, src_start, src_T, dT, 3, p[f"cw{i}"], p[f"cb{i}"], Cc)         # interpolate (i = 0) + conv
            ops.groupnorm_mish_(h, p[f"g{i}"], p[f"b{i}"], dstart, dT)
            src_start, src_T = dstart, dT
        h = ops.linear(h, p["ow"], p["ob"], Cc)                                                            # model[-1]: 1x1 conv
        o = 0
        for b in range(B):
            n_b = min(yl[b], fl[b])
            out[b, :n_b] = h[o:o + n_b]                                                                    # out * mask
            o += fl[b]
        return out, torch.as_tensor(yl, device=dev), None, None, None

    __call__ = forward
