"""Host-side mirror of the reference's s2mel flow-matching decoder (`CFM` = `BASECFM` around the `DiT` estimator), backed by
the HIP engine.

Reference interface mirrored (indextts/s2mel/modules/flow_matching.py): `CFM(args)`, `.load_state_dict`, `.estimator(...)`
(= `DiT.forward`, diffusion_transformer.py:186-257), `.inference(mu, x_lens, prompt, style, f0, n_timesteps, temperature=1.0,
inference_cfg_rate=0.5)` (:30-55) and `.solve_euler(x, x_lens, prompt, mu, style, f0, t_span, inference_cfg_rate)` (:57-115);
call site indextts/infer_v2_5.py:841-845.

What runs where: the 25-step CFG Euler loop and the whole estimator (13 transformer layers with adaptive RMSNorm / RoPE /
SwiGLU / U-ViT skips, non-causal attention, the WaveNet head) run inside `libindextts_hip.so` (`itts_s2mel_solve`).  Host-side
torch computes only vectors that depend on the timestep alone (timestep embeddings -> AdaLN / WaveNet-conditioning /
final-layer modulation vectors, one small matrix-vector product each per step) and, once per call, the step-invariant part of
`cond_x_merge_linear` through the engine's own GEMM entry point.

New capability (the reference runs this stage at batch 1, infer_v2_5.py:201): any number of utterances in one call, each with
its own length and prompt length; rows are packed, so there is no padding work.  `frame_lens` selects how many frames of a row
are processed (default: the tensor width for every row -- the reference's semantics, where frames past `x_lens` are still
computed and the WaveNet's reflect padding sits at the tensor end; the pipeline passes `frame_lens = x_lens`, which is what a
batch-1 reference call per utterance does).
"""
import ctypes as C
import os
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import _lib
from .bigvgan import fold_weight_norm
from .gpt import gemm as engine_gemm
from .gpt import linear_f32, pack_gemm_weight


def _get(obj, *path, default=None):
    """attribute-or-key access along a path (the reference passes Munch / OmegaConf objects, tests pass dicts)"""
    cur = obj
    for p in path:
        if cur is None:
            return default
        if isinstance(cur, dict):
            cur = cur.get(p, None)
        else:
            cur = getattr(cur, p, None)
    return default if cur is None else cur


class CFM:
    def __init__(self, args, precision: str = "bf16", device="cuda:0"):
        """`args` = the reference's `cfg.s2mel` mapping (DiT.*, wavenet.*, style_encoder.dim)."""
        self.hidden_dim = int(_get(args, "DiT", "hidden_dim"))
        self.num_heads = int(_get(args, "DiT", "num_heads"))
        self.depth = int(_get(args, "DiT", "depth"))
        self.in_channels = int(_get(args, "DiT", "in_channels"))
        self.content_dim = int(_get(args, "DiT", "content_dim"))
        self.style_dim = int(_get(args, "style_encoder", "dim"))
        self.wavenet_hidden = int(_get(args, "wavenet", "hidden_dim"))
        self.wavenet_layers = int(_get(args, "wavenet", "num_layers"))
        self.wavenet_kernel = int(_get(args, "wavenet", "kernel_size"))
        self.wavenet_dilation_rate = int(_get(args, "wavenet", "dilation_rate"))
        for flag, want in (("long_skip_connection", True), ("uvit_skip_connection", True), ("style_condition", True),
                           ("is_causal", False), ("time_as_token", False), ("style_as_token", False)):
            have = _get(args, "DiT", flag, default=want)
            if bool(have) != want:
                raise NotImplementedError(f"DiT.{flag}={have}: the engine implements the shipped IndexTTS-2 configuration ({want})")
        if str(_get(args, "DiT", "final_layer_type", default="wavenet")) != "wavenet":
            raise NotImplementedError("only final_layer_type='wavenet' is used by the IndexTTS checkpoints")
        self.zero_prompt_speech_token = bool(_get(args, "DiT", "zero_prompt_speech_token", default=False))
        self.rope_base = 10000.0
        self.norm_eps = 1e-5
        self.device = torch.device(device)
        # "fp32x3": f32 activations, GEMMs on the bf16 matrix pipe with every f32 operand carried exactly as three bf16 planes
        # (gemm_x3_kernel); attention, norms and every element-wise stage run their f32 code
        self.precision = {"bf16": 1, "bfloat16": 1, "fp32": 0, "float32": 0, "f32": 0, "fp32x3": 2, "f32x3": 2}[precision]
        # the once-per-solve projections of the step-invariant inputs (K = 784 is not a multiple of the x3 kernel's 32-deep tile) run f32
        self._host_prec = 0 if self.precision == 2 else self.precision
        # solve_euler: skip the post-attention stages on prompt rows whose output the Euler step discards (bit-identical results; A/B switch)
        self.prune_dead_rows = True
        cfg = _lib.S2MelConfig()
        cfg.hidden_dim, cfg.num_heads, cfg.depth, cfg.in_channels = self.hidden_dim, self.num_heads, self.depth, self.in_channels
        cfg.wavenet_hidden, cfg.wavenet_layers = self.wavenet_hidden, self.wavenet_layers
        cfg.wavenet_kernel, cfg.wavenet_dilation_rate = self.wavenet_kernel, self.wavenet_dilation_rate
        cfg.precision, cfg.norm_eps = self.precision, self.norm_eps
        self._h = C.c_void_p()
        with _lib.on_device(self.device):
            _lib.check(_lib.lib().itts_s2mel_create(C.byref(cfg), C.byref(self._h)), "itts_s2mel_create")
        self._p: Dict[str, torch.Tensor] = {}
        self._lin_packed: Dict[str, torch.Tensor] = {}         # packed weights of the host-side projections, by parameter name (`_linear`)
        self._loaded = False
        self._ws = None
        self.estimator = self._estimator_call

    # ---- checkpoint ------------------------------------------------------------------------------------------
    _ENGINE = ("attention.wqkv.weight", "attention.wo.weight", "feed_forward.w1.weight", "feed_forward.w2.weight",
               "feed_forward.w3.weight", "attention_norm.norm.weight", "ffn_norm.norm.weight", "skip_in_linear.weight",
               "skip_in_linear.bias")
    _ENGINE_TOP = ("transformer.norm.norm.weight", "cond_x_merge_linear.weight", "skip_linear.weight", "skip_linear.bias",
                   "conv1.weight", "conv1.bias", "res_projection.weight", "res_projection.bias", "final_layer.linear.weight",
                   "final_layer.linear.bias", "conv2.weight", "conv2.bias")

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        """Reference `s2mel.pth` names for the `cfm` model (`estimator.*`; weight-norm tensors are folded here)."""
        L = _lib.lib()
        sd = fold_weight_norm({k: v for k, v in sd.items()})
        self._lin_packed.clear()
        P = "estimator."
        ignored = []
        with _lib.on_device(self.device):
            for name, t in sd.items():
                if not name.startswith(P):
                    ignored.append(name)
                    continue
                tail = name[len(P):]
                to_engine = tail in self._ENGINE_TOP or (tail.startswith("transformer.layers.") and tail.split(".", 3)[3] in self._ENGINE) \
                    or (tail.startswith("wavenet.in_layers.") or tail.startswith("wavenet.res_skip_layers."))
                if to_engine:
                    tt = t.detach().to("cpu", torch.float32).contiguous()
                    if tail in ("conv2.weight",) and tt.dim() == 3:
                        tt = tt.reshape(tt.shape[0], tt.shape[1]).contiguous()        # Conv1d k=1 == Linear
                    shape = (C.c_int64 * tt.dim())(*tt.shape)
                    _lib.check(L.itts_s2mel_load_tensor(self._h, name.encode(), C.c_void_p(tt.data_ptr()), shape, tt.dim()),
                               f"itts_s2mel_load_tensor({name})")
                # host-side copies: everything that acts on vectors / the step-invariant merge columns
                if not to_engine or tail == "cond_x_merge_linear.weight":
                    self._p[tail] = t.detach().to(self.device, torch.float32).contiguous()
            _lib.check(L.itts_s2mel_finalize(self._h), "itts_s2mel_finalize")
        need = ["t_embedder.freqs", "t_embedder.mlp.0.weight", "t_embedder.mlp.2.weight", "t_embedder2.mlp.0.weight",
                "cond_projection.weight", "cond_x_merge_linear.weight", "cond_x_merge_linear.bias",
                "wavenet.cond_layer.conv.conv.weight", "final_layer.adaLN_modulation.1.weight",
                "transformer.norm.project_layer.weight"]
        missing = [n for n in need if n not in self._p]
        if missing:
            raise _lib.HipEngineError(f"CFM.load_state_dict: missing {missing}")
        # step-invariant merge: columns [C : C + C + H + style) of cond_x_merge_linear, as one packed engine GEMM
        Cc, H = self.in_channels, self.hidden_dim
        wm = self._p["cond_x_merge_linear.weight"]
        k_rest = wm.shape[1] - Cc
        self._k_rest = k_rest
        self._k_rest_pad = (k_rest + 63) // 64 * 64
        w_rest = torch.zeros(self._k_rest_pad, H)
        w_rest[:k_rest] = wm[:, Cc:].t().cpu()
        self._w_rest = pack_gemm_weight(w_rest, self._host_prec).to(self.device)
        cd = self.content_dim
        self._cd_pad = (cd + 63) // 64 * 64
        w_cp = torch.zeros(self._cd_pad, H)
        w_cp[:cd] = self._p["cond_projection.weight"].t().cpu()
        self._w_cp = pack_gemm_weight(w_cp, self._host_prec).to(self.device)
        self._loaded = True
        return ignored

    def eval(self):
        return self

    def to(self, device):
        d = torch.device(device)
        if d.type == "cuda" and d.index is not None and self.device.index is not None and d.index != self.device.index:
            raise _lib.HipEngineError(f"CFM was built on {self.device}; construct it with device={device!r} instead")
        return self

    # ---- per-step vectors (host torch, f32) ------------------------------------------------------------------------
    def _linear(self, x: torch.Tensor, name: str) -> torch.Tensor:
        """`F.linear(x, p[name + ".weight"], p[name + ".bias"])` on the engine's f32 GEMM; the packed weight is kept on THIS model under the
        parameter's name (dropped by load_state_dict)"""
        w = self._p[name + ".weight"]
        return linear_f32(x, w.reshape(w.shape[0], -1), self._p[name + ".bias"], cache=self._lin_packed, key=name)


    def _timestep_embed(self, prefix: str, t: torch.Tensor) -> torch.Tensor:      # diffusion_transformer.py:20-60
        p = self._p
        args = 1000 * t[:, None].float() * p[prefix + "freqs"][None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        h = F.silu(self._linear(emb, prefix + "mlp.0"))
        return self._linear(h, prefix + "mlp.2")

    def _mods(self, t: torch.Tensor) -> torch.Tensor:
        """t (n_steps,) -> (n_steps, mods_per_step) in the order itts_s2mel_mods_per_step documents."""
        p = self._p
        t1 = self._timestep_embed("t_embedder.", t)
        t2 = self._timestep_embed("t_embedder2.", t)
        parts = []
        for i in range(self.depth):
            L = f"transformer.layers.{i}."
            for nm in ("attention_norm.", "ffn_norm."):
                parts.append(self._linear(t1, L + nm + "project_layer"))
        parts.append(self._linear(t1, "transformer.norm.project_layer"))
        parts.append(self._linear(t2, "wavenet.cond_layer.conv.conv"))                       # (a k = 1 conv: its weight as (out, in))
        parts.append(self._linear(F.silu(t1), "final_layer.adaLN_modulation.1"))
        out = torch.cat(parts, dim=-1).contiguous()
        assert out.shape[1] == _lib.lib().itts_s2mel_mods_per_step(self._h), out.shape
        return out

    def _rope(self, T: int) -> torch.Tensor:                                      # gpt_fast/model.py:336-345
        n = 64
        freqs = 1.0 / (self.rope_base ** (torch.arange(0, n, 2, device=self.device)[: n // 2].float() / n))
        ang = torch.outer(torch.arange(T, device=self.device).float(), freqs)
        return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).contiguous()

    def _act(self, x: torch.Tensor) -> torch.Tensor:
        return x.bfloat16().contiguous() if self.precision == 1 else x.float().contiguous()

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    # ---- packing -------------------------------------------------------------------------------------------------
    def _tables(self, frame_lens, x_lens, n_branch):
        """Sequence tables for `n_branch` copies of the batch (branch-major), built ON THE DEVICE from the per-sequence lengths (a few hundred bytes
        cross to the device; the 2 x n_tok row tables -- 2.5 MB at 64 utterances -- used to be built by host torch ops and copied: ~100 ms of
        idle device per 64-utterance solve, profiles/r06f/gap_report)."""
        dev = self.device
        fl = torch.as_tensor(frame_lens, dtype=torch.int32).reshape(-1)
        xl = torch.minimum(torch.as_tensor(x_lens, dtype=torch.int32).reshape(-1), fl)
        n_tok, t_max = int(fl.sum()) * n_branch, int(fl.max())
        seq_T = fl.repeat(n_branch).to(dev)
        seq_len = xl.repeat(n_branch).to(dev)
        seq_start = torch.cumsum(seq_T, 0, dtype=torch.int32) - seq_T
        tok_seq = torch.repeat_interleave(torch.arange(seq_T.numel(), dtype=torch.int32, device=dev), seq_T.long(), output_size=n_tok)
        tok_t = torch.arange(n_tok, dtype=torch.int32, device=dev) - seq_start[tok_seq.long()]
        t = dict(seq_T=seq_T, seq_len=seq_len, seq_start=seq_start, tok_seq=tok_seq, tok_t=tok_t)
        return {k: v.contiguous() for k, v in t.items()}, n_tok, t_max

    def _tail_tables(self, frame_lens, x_lens, prompt_lens, n_branch):
        """Tables of the TAIL layout for the solver's dead-row elimination (itts_s2mel_set_tail): of every sequence only the frames
        from `prompt_len - halo` on, halo = the WaveNet stack's one-sided receptive field (sum of (k - 1) / 2 * dilation per layer).  The
        Euler step never reads the estimator at prompt frames (flow_matching.py:107) and the stages after the last attention are
        row-wise apart from that halo, so their outputs at the kept target frames are bit-identical.  None when nothing can be cut.
        Built on the device like `_tables`."""
        halo = sum((self.wavenet_kernel - 1) // 2 * self.wavenet_dilation_rate ** i for i in range(self.wavenet_layers))
        fl = [int(v) for v in frame_lens]
        xl = [min(int(v), f) for v, f in zip(torch.as_tensor(x_lens).reshape(-1).tolist(), fl)]
        cuts = [max(0, min(int(p), f) - halo) for p, f in zip(prompt_lens, fl)]
        span = (self.wavenet_kernel - 1) * self.wavenet_dilation_rate ** (self.wavenet_layers - 1)
        if not any(cuts) or any(f - c <= span for f, c in zip(fl, cuts)):          # nothing to cut / a tail shorter than one conv's padding
            return None
        dev = self.device
        fl2 = [f - c for f, c in zip(fl, cuts)]
        xl2 = [max(0, x - c) for x, c in zip(xl, cuts)]
        n_tok2 = sum(fl2) * n_branch
        full_T = torch.tensor(fl * n_branch, dtype=torch.int64).to(dev)
        full_start = torch.cumsum(full_T, 0) - full_T
        T2 = torch.tensor(fl2 * n_branch, dtype=torch.int64).to(dev)
        start2 = torch.cumsum(T2, 0) - T2
        cut = torch.tensor(cuts * n_branch, dtype=torch.int64).to(dev)
        tok_seq = torch.repeat_interleave(torch.arange(T2.numel(), device=dev), T2, output_size=n_tok2)
        tok_t = torch.arange(n_tok2, device=dev) - start2[tok_seq]
        t = dict(tok_seq=tok_seq, tok_t=tok_t, seq_start=start2, seq_T=T2, seq_len=torch.tensor(xl2 * n_branch).to(dev),
                 tail_src=full_start[tok_seq] + cut[tok_seq] + tok_t, tail_base=start2 - cut)
        out = {k: v.to(torch.int32).contiguous() for k, v in t.items()}
        out["n_tok"], out["t_max"] = n_tok2, int(max(fl2))
        self._tail_keep = out                                       # the engine reads the arrays during the solve
        return out

    def _const_in(self, prompt_x_rows, mu_rows, style_rows, n_null_rows):
        """cond_x_merge_linear on the step-invariant columns [prompt | cond_projection(mu) | style] + bias for the conditional
        rows; the null branch (all three inputs zero) sees cond_projection(0) = its bias.  Engine GEMMs."""
        p, H = self._p, self.hidden_dim
        n = mu_rows.shape[0]
        a = torch.zeros(n, self._cd_pad, device=self.device)
        a[:, : self.content_dim] = mu_rows
        cond = engine_gemm(self._act(a), self._w_cp, p["cond_projection.bias"], H, self._host_prec, prefill_tiles=True)
        rest = torch.zeros(n, self._k_rest_pad, device=self.device)
        Cc = self.in_channels
        rest[:, :Cc] = prompt_x_rows
        rest[:, Cc:Cc + H] = cond
        rest[:, Cc + H:Cc + H + self.style_dim] = style_rows
        cin = engine_gemm(self._act(rest), self._w_rest, p["cond_x_merge_linear.bias"], H, self._host_prec, prefill_tiles=True)
        if n_null_rows:
            null = torch.zeros(1, self._k_rest_pad, device=self.device)
            null[:, Cc:Cc + H] = p["cond_projection.bias"]
            nrow = engine_gemm(self._act(null), self._w_rest, p["cond_x_merge_linear.bias"], H, self._host_prec, prefill_tiles=True)
            cin = torch.cat([cin, nrow.expand(n_null_rows, -1)], 0)
        return cin.contiguous()

    @staticmethod
    def _pack_rows(x_bct: torch.Tensor, seq: torch.Tensor, frame: torch.Tensor) -> torch.Tensor:
        """(B, C, T) -> packed (n_rows, C): row i = frame frame[i] of sequence seq[i] (one gather through the row tables of the first branch)"""
        return x_bct.transpose(1, 2)[seq, frame].contiguous()

    @staticmethod
    def _unpack_rows(rows: torch.Tensor, seq: torch.Tensor, frame: torch.Tensor, B: int, T: int) -> torch.Tensor:
        """packed (n_rows, C) -> (B, C, T), frames beyond a sequence's rows left at 0 (one scatter)"""
        out = torch.zeros(B, T, rows.shape[1], device=rows.device, dtype=rows.dtype)
        out[seq, frame] = rows
        return out.transpose(1, 2).contiguous()

    # ---- DiT.forward (one estimator call; used by the parity tests) -----------------------------------------------
    def _estimator_call(self, x, prompt_x, x_lens, t, style, cond, frame_lens=None):
        """x, prompt_x (B, C, T); x_lens (B,) or (B/2,) broadcast over a CFG-stacked batch; t (B,); style (B, style_dim);
        cond (B, T, content_dim) -> (B, C, T), rows beyond a row's frame_lens left at 0."""
        if not self._loaded:
            raise RuntimeError("CFM: load_state_dict() first")
        dev = self.device
        B, Cc, T = x.shape
        x_lens = torch.as_tensor(x_lens).reshape(-1)
        if x_lens.numel() != B:
            x_lens = x_lens.repeat(B // x_lens.numel())
        fl = [T] * B if frame_lens is None else [int(v) for v in frame_lens]
        tabs, n_tok, t_max = self._tables(fl, x_lens, 1)
        with _lib.on_device(dev):
            sq, fr = tabs["tok_seq"].long(), tabs["tok_t"].long()
            xs = self._pack_rows(x.to(dev).float(), sq, fr)
            cin = self._const_in(self._pack_rows(prompt_x.to(dev).float(), sq, fr), cond.to(dev).float()[sq, fr], style.to(dev).float()[sq], 0)
            tt = torch.as_tensor(t, dtype=torch.float32).reshape(-1).to(dev)
            if not bool((tt == tt[0]).all()):
                raise NotImplementedError("one timestep per call (the solver never mixes timesteps in a batch)")
            mods = self._mods(tt[:1])
            rope = self._rope(t_max)
            L = _lib.lib()
            ws = self._workspace(L.itts_s2mel_workspace_bytes(self._h, n_tok, B, t_max))
            d = torch.empty(n_tok, Cc, dtype=torch.float32, device=dev)
            _lib.check(L.itts_s2mel_estimator(self._h, _lib.ptr(xs), _lib.ptr(cin), _lib.ptr(mods), _lib.ptr(rope),
                                              _lib.ptr(tabs["tok_seq"]), _lib.ptr(tabs["tok_t"]), _lib.ptr(tabs["seq_start"]),
                                              _lib.ptr(tabs["seq_T"]), _lib.ptr(tabs["seq_len"]), B, n_tok, t_max, _lib.ptr(d),
                                              _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "itts_s2mel_estimator")
            out = self._unpack_rows(d, sq, fr, B, T)
        return out

    # ---- BASECFM.inference / solve_euler ----------------------------------------------------------------------------
    @torch.no_grad()
    def inference(self, mu, x_lens, prompt, style, f0, n_timesteps, temperature=1.0, inference_cfg_rate=0.5, noise=None,
                  prompt_lens=None, frame_lens=None):
        B, T = mu.size(0), mu.size(1)
        if noise is None:
            noise = torch.randn([B, self.in_channels, T], device=mu.device) * temperature
        t_span = torch.linspace(0, 1, n_timesteps + 1)
        return self.solve_euler(noise, x_lens, prompt, mu, style, f0, t_span, inference_cfg_rate, prompt_lens=prompt_lens,
                                frame_lens=frame_lens)

    @torch.no_grad()
    def solve_euler(self, x, x_lens, prompt, mu, style, f0, t_span, inference_cfg_rate=0.5, prompt_lens=None, frame_lens=None):
        """flow_matching.py:57-115.  x (B, C, T) noise; prompt (B or 1, C, Tp); mu (B, T, content_dim); style (B or 1, style_dim);
        prompt_lens (B,) when the prompts of a batch differ in length (default: prompt.size(-1) for every row)."""
        if not self._loaded:
            raise RuntimeError("CFM: load_state_dict() first")
        if f0 is not None:
            raise NotImplementedError("f0 conditioning is not used by the IndexTTS-2 s2mel checkpoint")
        dev = self.device
        B, Cc, T = x.shape
        x_lens = torch.as_tensor(x_lens).reshape(-1)
        fl = [T] * B if frame_lens is None else [int(v) for v in frame_lens]
        pl = [int(prompt.size(-1))] * B if prompt_lens is None else [int(v) for v in prompt_lens]
        nb = 2 if inference_cfg_rate > 0 else 1
        tabs, n_tok, t_max = self._tables(fl, x_lens, nb)
        n_steps = int(t_span.numel()) - 1
        with _lib.on_device(dev):
            x = x.to(dev).float()
            prompt = prompt.to(dev).float()
            if prompt.shape[0] == 1 and B > 1:
                prompt = prompt.expand(B, -1, -1)
            if style.shape[0] == 1 and B > 1:
                style = style.expand(B, -1)
            # flow_matching.py:84-89 for every row at once: prompt frames of x are zeroed, prompt_x holds the prompt there and zeros elsewhere
            Tp = int(prompt.size(-1))
            in_prompt = torch.arange(T, device=dev)[None, :] < torch.tensor(pl, device=dev)[:, None]            # (B, T)
            prompt_x = torch.zeros_like(x)
            prompt_x[:, :, :Tp] = prompt * in_prompt[:, None, :Tp]
            x = torch.where(in_prompt[:, None, :], torch.zeros((), device=dev), x)
            mu = mu.to(dev).float()
            if self.zero_prompt_speech_token:
                mu = torch.where(in_prompt[:, :, None], torch.zeros((), device=dev), mu)
            n_rows = n_tok // nb
            sq, fr = tabs["tok_seq"][:n_rows].long(), tabs["tok_t"][:n_rows].long()                             # the first branch's rows
            xs = self._pack_rows(x, sq, fr)
            cin = self._const_in(self._pack_rows(prompt_x, sq, fr), mu[sq, fr], style.to(dev).float()[sq], n_rows if nb == 2 else 0)
            # t accumulates in fp32 exactly as the reference loop does (t = t + dt)
            ts = t_span.detach().to("cpu", torch.float32)
            t_list, t = [], ts[0].clone()
            for k in range(1, n_steps + 1):
                t_list.append(t.clone())
                t = t + (ts[k] - ts[k - 1])
            mods = self._mods(torch.stack(t_list).to(dev))
            rope = self._rope(t_max)
            plen = torch.tensor(pl * nb, dtype=torch.int32, device=dev)
            L = _lib.lib()
            ws = self._workspace(L.itts_s2mel_workspace_bytes(self._h, n_tok, B * nb, t_max))
            tsp = (C.c_float * (n_steps + 1))(*[float(v) for v in ts])
            tail = self._tail_tables(fl, x_lens, pl, nb) if self.prune_dead_rows else None
            if tail is not None:
                _lib.check(L.itts_s2mel_set_tail(self._h, *[_lib.ptr(tail[k]) for k in ("tok_seq", "tok_t", "seq_start", "seq_T", "seq_len",
                                                                                        "tail_src", "tail_base")],
                                                 B * nb, tail["n_tok"], tail["t_max"]), "itts_s2mel_set_tail")
            else:
                L.itts_s2mel_set_tail(self._h, None, None, None, None, None, None, None, 0, 0, 0)
            _lib.check(L.itts_s2mel_solve(self._h, _lib.ptr(xs), _lib.ptr(cin), _lib.ptr(mods), _lib.ptr(rope), _lib.ptr(tabs["tok_seq"]),
                                          _lib.ptr(tabs["tok_t"]), _lib.ptr(tabs["seq_start"]), _lib.ptr(tabs["seq_T"]),
                                          _lib.ptr(tabs["seq_len"]), _lib.ptr(plen), B * nb, n_tok, t_max, nb, n_steps, tsp,
                                          float(inference_cfg_rate), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "itts_s2mel_solve")
            out = self._unpack_rows(xs, sq, fr, B, T)
        return out

    # ---- HIP-event profile of the last solve (bench.py) -----------------------------------------------------------
    def set_profiling(self, enable: bool):
        _lib.check(_lib.lib().itts_s2mel_set_profiling(self._h, int(enable)), "itts_s2mel_set_profiling")

    def profile(self):
        """{gemm: {ms, launches, flops}, attention: {ms, launches}, estimator_calls: {ms, launches}} of the last solve."""
        arr = [(C.c_double * 3)() for _ in range(3)]
        _lib.check(_lib.lib().itts_s2mel_profile_read(self._h, *arr), "itts_s2mel_profile_read")
        names = ("gemm", "attention", "estimator_calls")
        return {n: dict(ms=arr[0][i], launches=int(arr[1][i]), flops=arr[2][i]) for i, n in enumerate(names)}

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _lib.lib().itts_s2mel_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


def codes_to_mel(semantic_codec, s2mel_models, codes: torch.Tensor, code_lens, bundle, duration_factor: float = 1.0,
                 diffusion_steps: int = 25, inference_cfg_rate: float = 0.7, noise: Optional[torch.Tensor] = None):
    """indextts/infer_v2_5.py:830-846 for a whole batch of segments on the HIP engine: semantic_codec.decode -> length_regulator
    -> [prompt_condition | cond] -> cfm.inference -> drop the prompt frames.  Every row is processed at its own lengths (what the
    reference's batch-1 call per segment computes).  bundle: prompt_condition (1, Tp, 512), ref_mel (1, 80, Tp), style (1, 192).
    Returns mel (B, 80, max frames) f32 and the frame counts (B,) int32."""
    lens = [int(v) for v in code_lens]
    S_infer = semantic_codec.decode(codes, code_lens=lens)                                     # (B, 2T, 1024)
    target = [int(2 * n * 1.72 * duration_factor) for n in lens]                               # :833
    reg, cfm = s2mel_models["length_regulator"], s2mel_models["cfm"]
    cond = reg(S_infer, ylens=torch.tensor(target), n_quantizers=3, f0=None, xlens=[2 * n for n in lens], frame_lens=target)[0]
    prompt_condition, ref_mel, style = bundle["prompt_condition"], bundle["ref_mel"], bundle["style"]
    Tp = int(prompt_condition.shape[1])
    B = codes.shape[0]
    total = [Tp + t for t in target]
    cat = torch.zeros(B, max(total), cond.shape[-1], dtype=torch.float32, device=cond.device)
    cat[:, :Tp] = prompt_condition.to(cond.device, torch.float32)
    for b in range(B):
        cat[b, Tp:total[b]] = cond[b, : target[b]]
    mel = cfm.inference(cat, torch.tensor(total), ref_mel, style, None, diffusion_steps, inference_cfg_rate=inference_cfg_rate,
                        noise=noise, frame_lens=total)
    return mel[:, :, Tp:].contiguous(), torch.tensor(target, dtype=torch.int32)


class GptLayer:
    """`s2mel.models['gpt_layer']` of IndexTTS-2 (commons.py:413): Linear(1280, 256) -> Linear(256, 128) -> Linear(128, 1024)
    on the GPT latents, three engine GEMMs in f32."""

    def __init__(self, dims=(1280, 256, 128, 1024), device="cuda:0"):
        self.dims, self.device = tuple(dims), torch.device(device)
        self._w = []

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        self._w = []
        for i in range(len(self.dims) - 1):
            w, b = sd[f"{i}.weight"], sd[f"{i}.bias"]
            if tuple(w.shape) != (self.dims[i + 1], self.dims[i]):
                raise _lib.HipEngineError(f"gpt_layer.{i}.weight has shape {tuple(w.shape)}, expected {(self.dims[i + 1], self.dims[i])}")
            self._w.append((pack_gemm_weight(w.detach().float().cpu().t().contiguous(), 0).to(self.device),
                            b.detach().to(self.device, torch.float32).contiguous(), self.dims[i + 1]))
        return []

    @torch.no_grad()
    def __call__(self, latent: torch.Tensor) -> torch.Tensor:
        shape = latent.shape
        x = latent.reshape(-1, shape[-1]).to(self.device, torch.float32).contiguous()
        for wp, b, n in self._w:
            x = engine_gemm(x, wp, b, n, 0, prefill_tiles=True)
        return x.reshape(*shape[:-1], self.dims[-1])


class MyModel:
    """`indextts/s2mel/modules/commons.py::MyModel` as the pipeline uses it (infer_v2_5.py:190-206): a `.models` mapping with
    the flow-matching decoder and the length regulator, both on the HIP engine."""

    def __init__(self, args, use_gpt_latent: bool = False, precision: str = "bf16", device="cuda:0"):
        from .codec import InterpolateRegulator
        lr = _get(args, "length_regulator")
        self.models = {
            "cfm": CFM(args, precision=precision, device=device),
            "length_regulator": InterpolateRegulator(
                channels=int(_get(lr, "channels")), sampling_ratios=tuple(_get(lr, "sampling_ratios")),
                is_discrete=bool(_get(lr, "is_discrete", default=False)), in_channels=_get(lr, "in_channels"),
                vector_quantize=bool(_get(lr, "vector_quantize", default=False)),
                codebook_size=int(_get(lr, "content_codebook_size", default=1024)),
                f0_condition=bool(_get(lr, "f0_condition", default=False)), device=device),
        }
        if use_gpt_latent:                                            # IndexTTS-2 (infer_v2.py:101): latent projector
            self.models["gpt_layer"] = GptLayer(device=device)

    def load_state_dict(self, net: Dict[str, Dict[str, torch.Tensor]]):
        """`net` = the checkpoint's `state['net']` mapping (load_checkpoint2, commons.py): {'cfm': sd, 'length_regulator': sd};
        `module.` prefixes of DDP checkpoints are stripped."""
        strip = lambda sd: {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.models["cfm"].load_state_dict(strip(net["cfm"]))
        self.models["length_regulator"].load_state_dict(strip(net["length_regulator"]))
        if "gpt_layer" in self.models:
            self.models["gpt_layer"].load_state_dict(strip(net["gpt_layer"]))
        return self

    def eval(self):
        return self

    def to(self, device):
        return self
