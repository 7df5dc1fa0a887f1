"""Host-side mirror of the reference BigVGAN vocoder classes, backed by the HIP engine.

Reference interface mirrored (same names / argument meaning):
  * v2 / v2.5: `indextts/s2mel/modules/bigvgan/bigvgan.py::BigVGAN` -- `BigVGAN(h, use_cuda_kernel)`,
    `BigVGAN.from_pretrained(dir)`, `.remove_weight_norm()`, `.eval()`, `.to(device)`, `model(mel) -> (B,1,T*256)`
    (call site `indextts/infer_v2_5.py:224-233,850`).
  * v1 / v1.5: `indextts/BigVGAN/models.py::BigVGAN` -- `model(latent (B,T,D), mel_ref^T) -> (wav, None)`
    (call site `indextts/infer.py:647`).  The ECAPA-TDNN speaker encoder runs on the engine too (indextts_amd/ecapa.py, built from the
    checkpoint's `speaker_encoder.*` tensors; `speaker_encoder=` injects another callable); its embedding is cached per reference clip instead
    of being recomputed per call.
Extra (not in the reference): `lens=` for ragged batches -- every row is bounded at its own length so the result
per row equals the reference run at B=1.
"""
import ctypes as C
import json
import os
from typing import Dict, Optional

import torch

from . import _lib

_V2_DEFAULTS = dict(
    num_mels=80, upsample_rates=[4, 4, 2, 2, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4],
    upsample_initial_channel=1536, resblock_kernel_sizes=[3, 7, 11],
    resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], activation="snakebeta", snake_logscale=True,
    use_tanh_at_final=True, use_bias_at_final=True, resblock="1")


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """`remove_weight_norm()` at load time: weight = g * v / ||v|| (norm over all dims but 0)."""
    out = dict(sd)
    for k in list(sd.keys()):
        if k.endswith(".weight_g"):
            base, g, v = k[:-9], sd[k], sd[k[:-9] + ".weight_v"]
        elif k.endswith(".parametrizations.weight.original0"):
            base = k[: -len(".parametrizations.weight.original0")]
            g, v = sd[k], sd[base + ".parametrizations.weight.original1"]
        else:
            continue
        dims = tuple(range(1, v.dim()))
        out[base + ".weight"] = v * (g / v.norm(2, dim=dims, keepdim=True))
        for suf in (".weight_g", ".weight_v", ".parametrizations.weight.original0",
                    ".parametrizations.weight.original1"):
            out.pop(base + suf, None)
    return out


class BigVGAN:
    """BigVGAN generator on the HIP engine.  `h` is the reference hparams mapping (config.json)."""

    def __init__(self, h, use_cuda_kernel: bool = False, cond_dim: int = 0, in_channels: Optional[int] = None,
                 cond_in_each_up_layer: bool = True, speaker_encoder=None, device=None, conv_mode: Optional[str] = None,
                 h3_min_channels: int = 0):
        """conv_mode: None / "f32" = f32 MFMA convs (the parity mode, default); "bf16x3" = the resblock convs with >= h3_min_channels
        channels on the bf16 matrix cores with every f32 operand carried exactly as three bf16 planes, six plane products (the fp32x3
        arithmetic of the flow-matching stage: error vs f64 not above the f32 MFMA kernel's); "f16x3" = the opt-in 22-bit split-operand mode
        (three f16 MFMA products per f32 product)."""
        if conv_mode not in (None, "f32", "f16x3", "bf16x3"):
            raise ValueError(f"BigVGAN: conv_mode must be 'f32', 'bf16x3' or 'f16x3', got {conv_mode!r}")
        self.conv_mode = {"f16x3": 1, "bf16x3": 2}.get(conv_mode, 0)
        self.h3_min_channels = int(h3_min_channels)
        hp = dict(_V2_DEFAULTS)
        hp.update(dict(h))
        if str(hp.get("resblock", "1")) != "1":
            raise NotImplementedError("only AMPBlock1 (resblock='1') is used by the IndexTTS checkpoints")
        if hp["activation"] not in ("snakebeta", "snake"):
            raise NotImplementedError("activation incorrectly specified. check the config file and look for 'activation'.")
        self.h = hp
        self.use_cuda_kernel = use_cuda_kernel        # accepted for signature parity; the HIP kernels are always used
        self.cond_dim = int(cond_dim)
        self.speaker_encoder = speaker_encoder
        self._spk_cache = {}
        self.num_upsamples = len(hp["upsample_rates"])
        self.num_kernels = len(hp["resblock_kernel_sizes"])
        self.total_up = 1
        for u in hp["upsample_rates"]:
            self.total_up *= int(u)
        cfg = _lib.BigVGANConfig()
        cfg.in_channels = int(in_channels if in_channels is not None else hp["num_mels"])
        cfg.upsample_initial_channel = int(hp["upsample_initial_channel"])
        cfg.num_upsamples = self.num_upsamples
        for i, (u, k) in enumerate(zip(hp["upsample_rates"], hp["upsample_kernel_sizes"])):
            cfg.upsample_rates[i], cfg.upsample_kernel_sizes[i] = int(u), int(k)
        cfg.num_kernels = self.num_kernels
        nd = len(hp["resblock_dilation_sizes"][0])
        cfg.num_dilations = nd
        for j, k in enumerate(hp["resblock_kernel_sizes"]):
            cfg.resblock_kernel_sizes[j] = int(k)
            if len(hp["resblock_dilation_sizes"][j]) != nd:
                raise ValueError("all resblocks must have the same number of dilations")
            for d, dv in enumerate(hp["resblock_dilation_sizes"][j]):
                cfg.resblock_dilations[j][d] = int(dv)
        cfg.snake_logscale = int(bool(hp.get("snake_logscale", True)))
        cfg.activation = 0 if hp["activation"] == "snakebeta" else 1
        cfg.use_tanh_at_final = int(bool(hp.get("use_tanh_at_final", True)))
        cfg.use_bias_at_final = int(bool(hp.get("use_bias_at_final", True)))
        cfg.cond_dim = self.cond_dim
        cfg.cond_in_each_up_layer = int(bool(cond_in_each_up_layer))
        self._cfg = cfg
        self.in_channels = cfg.in_channels
        self._h = C.c_void_p()
        self._loaded = False
        self._ws = None
        self._sd_host = None              # folded f32 CPU tensors, kept so .to(other_device) can rebuild the handle there
        self.device = torch.device(device) if device is not None else None
        self._create_handle()

    def _create_handle(self):
        """(Re)create the C handle on `self.device` (or the current device): the handle's weights and events belong to the
        device that is current at itts_bigvgan_create."""
        L = _lib.lib()
        if self._h.value:
            L.itts_bigvgan_destroy(self._h)
            self._h = C.c_void_p()
        with _lib.on_device(self.device):
            _lib.check(L.itts_bigvgan_create(C.byref(self._cfg), C.byref(self._h)), "itts_bigvgan_create")
            if self.conv_mode:
                _lib.check(L.itts_bigvgan_set_conv_mode(self._h, self.conv_mode, self.h3_min_channels), "itts_bigvgan_set_conv_mode")
        self._loaded = False
        self._ws = None

    def _upload(self, strict: bool):
        L = _lib.lib()
        skipped = []
        for name, t in self._sd_host.items():
            shape = (C.c_int64 * t.dim())(*t.shape)
            rc = L.itts_bigvgan_load_tensor(self._h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim())
            if rc != 0:
                if strict:
                    _lib.check(rc, f"itts_bigvgan_load_tensor({name})")
                skipped.append(name)
        _lib.check(L.itts_bigvgan_finalize(self._h), "itts_bigvgan_finalize")
        self._loaded = True
        return skipped

    # ---- checkpoint loading --------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """Reference state-dict names; weight-norm is folded here."""
        sd = fold_weight_norm({k: v for k, v in sd.items()})
        skipped, host = [], {}
        for name, t in sd.items():
            if name.startswith("speaker_encoder.") or name.endswith("num_batches_tracked") or name == "logit_scale":
                skipped.append(name)
                continue
            host[name] = t.detach().to("cpu", torch.float32).contiguous()
        self._sd_host = host
        # v1 / v1.5 checkpoints carry the ECAPA-TDNN speaker encoder under `speaker_encoder.` (models.py:191): kept on the host and built on the
        # engine (indextts_amd/ecapa.py) at the first speaker_embedding() call, unless a speaker_encoder= was injected
        self._spk_sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items()
                        if k.startswith("speaker_encoder.") and not k.endswith("num_batches_tracked")}
        return skipped + self._upload(strict)

    def _build_speaker_encoder(self):
        from .ecapa import ECAPA_TDNN
        sd, P = self._spk_sd, "speaker_encoder."
        w0 = sd[P + "blocks.0.conv.conv.weight"]
        C = int(w0.shape[0])
        scale = 1 + len({k.split(".")[5] for k in sd if k.startswith(P + "blocks.1.res2net_block.blocks.")})
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        enc = ECAPA_TDNN(int(w0.shape[1]), device=dev, lin_neurons=int(sd[P + "fc.conv.weight"].shape[0]), channels=[C] * 4 + [3 * C],
                         attention_channels=int(sd[P + "asp.tdnn.conv.conv.weight"].shape[0]),
                         se_channels=int(sd[P + "blocks.1.se_block.conv1.conv.weight"].shape[0]), res2net_scale=scale)
        return enc.load_state_dict(sd, prefix=P)

    @classmethod
    def from_pretrained(cls, model_dir: str, use_cuda_kernel: bool = False, **kw):
        """Loads `config.json` + `bigvgan_generator.pt` from a local directory (bigvgan.py:413-492, offline)."""
        with open(os.path.join(model_dir, "config.json")) as f:
            h = json.load(f)
        model = cls(h, use_cuda_kernel=use_cuda_kernel, **kw)
        ck = torch.load(os.path.join(model_dir, "bigvgan_generator.pt"), map_location="cpu")
        model.load_state_dict(ck["generator"] if "generator" in ck else ck)
        return model

    def remove_weight_norm(self):      # folded at load; kept for call-site compatibility
        return self

    def eval(self):
        return self

    def to(self, device):
        """`module.to(device)` of the reference (infer_v2_5.py:232): binds the engine to `device`.  If the weights were
        already uploaded to another GPU the handle is rebuilt on the new one from the retained host copy."""
        d = torch.device(device)
        if d.type != "cuda":
            raise _lib.HipEngineError("BigVGAN (HIP engine) has no CPU path")
        want = d.index if d.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", want)
        if _lib.lib().itts_bigvgan_device(self._h) != want:
            self._create_handle()
            if self._sd_host is not None:
                self._upload(strict=True)
        return self

    def float(self):
        return self

    # ---- forward -------------------------------------------------------------------------------------------
    def _workspace(self, B: int, T: int, device) -> torch.Tensor:
        need = _lib.lib().itts_bigvgan_workspace_bytes(self._h, B, T)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def speaker_embedding(self, mel_ref: torch.Tensor, lens=None, key=None) -> torch.Tensor:
        """v1: ECAPA-TDNN embedding of the reference mel (models.py:202), cached per `key`."""
        if self.speaker_encoder is None and getattr(self, "_spk_sd", None):
            self.speaker_encoder = self._build_speaker_encoder()
        if self.speaker_encoder is None:
            raise RuntimeError("this BigVGAN was built without a speaker_encoder and its checkpoint carried none; pass speaker_embedding=")
        if key is not None and key in self._spk_cache:
            return self._spk_cache[key]
        with torch.no_grad():
            e = self.speaker_encoder(mel_ref, lens)
        e = e.reshape(e.shape[0], -1).float().contiguous()
        if key is not None:
            self._spk_cache[key] = e
        return e

    def forward(self, x: torch.Tensor, mel_ref: Optional[torch.Tensor] = None, lens: Optional[torch.Tensor] = None,
                speaker_embedding: Optional[torch.Tensor] = None):
        if not self._loaded:
            raise RuntimeError("BigVGAN: load_state_dict() first")
        if not x.is_cuda:
            raise _lib.HipEngineError("BigVGAN (HIP engine) needs a CUDA/HIP tensor; there is no CPU path")
        v1 = self.cond_dim > 0
        if v1:
            x = x.transpose(1, 2)                       # latent (B,T,D) -> (B,D,T)  (models.py:222)
            if speaker_embedding is None:
                speaker_embedding = self.speaker_embedding(mel_ref)
            spk = speaker_embedding.reshape(-1, self.cond_dim).float().contiguous()
            if spk.shape[0] == 1 and x.shape[0] > 1:
                spk = spk.expand(x.shape[0], -1).contiguous()
        else:
            spk = None
        x = x.float().contiguous()
        B, Cin, T = x.shape
        if Cin != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {Cin}")
        wav = torch.empty(B, 1, T * self.total_up, dtype=torch.float32, device=x.device)
        if B == 0 or T == 0:
            return (wav, None) if v1 else wav
        lens_t = None
        if lens is not None:
            lens_t = torch.as_tensor(lens, dtype=torch.int32, device=x.device).contiguous()
            if lens_t.numel() != B:
                raise ValueError("lens must have one entry per batch row")
        ws = self._workspace(B, T, x.device)
        rc = _lib.lib().itts_bigvgan_forward(self._h, _lib.ptr(x), _lib.ptr(lens_t), _lib.ptr(spk), _lib.ptr(wav), B, T,
                                             _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device))
        _lib.check(rc, "itts_bigvgan_forward")
        if self.conv_mode == 1:                         # the f16 split mode cannot represent values outside the f16 range: fail, never garble (bf16 planes have the f32 range)
            bad = _lib.lib().itts_bigvgan_range_check(self._h)
            if bad:
                raise _lib.HipEngineError("BigVGAN(conv_mode='f16x3'): an activation was not finite or outside the f16 range "
                                          "(|x| >= 65504); use the exact f32 mode for this model / input")
        return (wav, None) if v1 else wav

    __call__ = forward

    # ---- streaming / chunked synthesis (BASELINE.json configs[4]: long-form, "streaming BigVGAN overlap-add") ----------
    # One-sided receptive field of the generator in mel frames: conv_pre 3 + sum over stages of the AMP-block reach
    # (96 samples at the stage rate for k=11, d=(1,3,5)) / cumulative upsampling + upsamplers.
    def receptive_field_frames(self) -> int:
        h = self.h
        kmax = max(h["resblock_kernel_sizes"])
        reach = 0
        for d in max(h["resblock_dilation_sizes"], key=lambda ds: sum(ds)):
            reach += 6 + (kmax - 1) // 2 * d + 6 + (kmax - 1) // 2
        frames, rate = 3.0, 1
        for u, k in zip(h["upsample_rates"], h["upsample_kernel_sizes"]):
            frames += (k / u) / rate              # transposed-conv taps reach k/u input samples
            rate *= u
            frames += reach / rate
        frames += (6 + 3) / rate                  # activation_post + conv_post
        return int(frames) + 2

    def stream(self, mel: torch.Tensor, chunk_frames: int = 256, halo_frames: Optional[int] = None):
        """Yield the waveform of `mel` (B, C, T) chunk by chunk.  Each chunk is synthesised with `halo_frames` of real
        context on both sides and the halo output is discarded (overlap-save), so the concatenation equals the
        one-shot `forward(mel)` -- unlike the reference's TensorRT streaming path, which re-synthesises overlapping
        code chunks and Hann-crossfades them (backends/trt/pipeline/streaming.py:57-68,140-172)."""
        halo = self.receptive_field_frames() if halo_frames is None else int(halo_frames)
        T = mel.shape[-1]
        up = self.total_up
        for t0 in range(0, T, chunk_frames):
            t1 = min(T, t0 + chunk_frames)
            a, b = max(0, t0 - halo), min(T, t1 + halo)
            w = self.forward(mel[..., a:b].contiguous())
            yield w[..., (t0 - a) * up: (t0 - a) * up + (t1 - t0) * up]

    def open_stream(self, chunk_frames: int = 256, halo_frames: Optional[int] = None) -> "BigVGANStream":
        """Push-style streaming for one utterance (`itts_bigvgan_stream_*`): feed mel chunks as they arrive, get the
        samples that became final; output trails the input by the halo until `push(..., last=True)`."""
        return BigVGANStream(self, chunk_frames, self.receptive_field_frames() if halo_frames is None else int(halo_frames))

    def forward_chunked(self, mel: torch.Tensor, chunk_frames: int = 256, halo_frames: Optional[int] = None) -> torch.Tensor:
        return torch.cat(list(self.stream(mel, chunk_frames, halo_frames)), dim=-1)

    def set_profiling(self, enable: bool):
        _lib.check(_lib.lib().itts_bigvgan_set_profiling(self._h, int(enable)), "itts_bigvgan_set_profiling")

    def profile_records(self):
        """Per-launch (class, ms, flops, bytes) of the last forward, in launch order."""
        buf = (C.c_double * (4 * 1024))()
        n = _lib.lib().itts_bigvgan_profile_records(self._h, buf, 1024)
        if n < 0:
            _lib.check(1, "itts_bigvgan_profile_records")
        return [(int(buf[4 * i]), buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(min(n, 1024))]

    def profile(self):
        """Per-kernel-class totals of the last forward (HIP events on the launch stream)."""
        arr = [(C.c_double * 4)() for _ in range(4)]
        _lib.check(_lib.lib().itts_bigvgan_profile_read(self._h, *arr), "itts_bigvgan_profile_read")
        names = ("conv1d_mfma", "conv_transpose1d_mfma", "aa_activation", "conv_post")
        return {n: dict(ms=arr[0][i], launches=int(arr[1][i]), flops=arr[2][i], bytes=arr[3][i]) for i, n in enumerate(names)}

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _lib.lib().itts_bigvgan_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class BigVGANStream:
    def __init__(self, model: BigVGAN, chunk_frames: int, halo_frames: int):
        if model.cond_dim > 0:
            raise NotImplementedError("streaming is wired for the v2 generator (no speaker conditioning)")
        self.model, self.chunk, self.halo = model, int(chunk_frames), int(halo_frames)
        self._s = C.c_void_p()
        _lib.check(_lib.lib().itts_bigvgan_stream_open(model._h, self.chunk, self.halo, C.byref(self._s)), "itts_bigvgan_stream_open")
        self._ws = None

    def push(self, mel_chunk: torch.Tensor, last: bool = False) -> torch.Tensor:
        """mel_chunk (C, n) or (1, C, n) with n <= chunk_frames (n = 0 allowed with last=True) -> (1, 1, samples)."""
        n_new = int(mel_chunk.shape[-1])
        x = mel_chunk.reshape(self.model.in_channels, n_new).float().contiguous()
        if not x.is_cuda:
            raise _lib.HipEngineError("BigVGANStream needs a CUDA/HIP tensor")
        L = _lib.lib()
        if self._ws is None:
            self._ws = torch.empty(L.itts_bigvgan_stream_workspace_bytes(self._s), dtype=torch.uint8, device=x.device)
        out = torch.empty((self.chunk + self.halo) * self.model.total_up, dtype=torch.float32, device=x.device)
        n = C.c_int32(0)
        _lib.check(L.itts_bigvgan_stream_push(self._s, _lib.ptr(x) if n_new else None, max(1, n_new), n_new,
                                              int(last), None, _lib.ptr(out), C.byref(n), _lib.ptr(self._ws), self._ws.numel(),
                                              _lib.stream_ptr(x.device)), "itts_bigvgan_stream_push")
        return out[: n.value].reshape(1, 1, -1)

    def close(self):
        if self._s and self._s.value:
            _lib.lib().itts_bigvgan_stream_close(self._s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- unit-level ops (1:1 with the reference's fused op and torch layers) ------------------------------------------
def anti_alias_activation(x, up_filter, down_filter, alpha, beta, lens=None, logscale=True):
    """Drop-in for `anti_alias_activation_cuda.forward(inputs, up_ftr, down_ftr, alpha, beta)`
    (alias_free_activation/cuda/activation1d.py:23-27)."""
    if not x.is_cuda:
        raise _lib.HipEngineError("anti_alias_activation needs a device tensor")
    x = x.float().contiguous()
    y = torch.empty_like(x)
    B, Cc, T = x.shape
    dev = x.device
    f = lambda t: t.detach().reshape(-1).to(dev, torch.float32).contiguous()
    a, b, fu, fd = f(alpha), f(beta), f(up_filter), f(down_filter)
    lens_t = None if lens is None else torch.as_tensor(lens, dtype=torch.int32, device=dev).contiguous()
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().itts_aa_act_forward(_lib.ptr(x), _lib.ptr(y), _lib.ptr(a), _lib.ptr(b), _lib.ptr(fu),
                                                  _lib.ptr(fd), B, Cc, T, _lib.ptr(lens_t), 1, int(logscale),
                                                  _lib.stream_ptr(x.device)), "itts_aa_act_forward")
        return y


def pack_conv1d_weight(w: torch.Tensor) -> torch.Tensor:
    w = w.detach().to("cpu", torch.float32).contiguous()
    Cout, Cin, k = w.shape
    L = _lib.lib()
    out = torch.empty(L.itts_packed_conv_floats(Cout, Cin, k), dtype=torch.float32)
    _lib.check(L.itts_pack_conv1d_weight(_lib.ptr(w), Cout, Cin, k, _lib.ptr(out)), "itts_pack_conv1d_weight")
    return out


def pack_conv1d_h3_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout][Cin][k] f32 -> the split f16 fragment streams of the f16 x 3 conv (uint8 host tensor)"""
    w = w.detach().to("cpu", torch.float32).contiguous()
    Cout, Cin, k = w.shape
    L = _lib.lib()
    n = L.itts_conv1d_h3_packed_bytes(Cout, Cin, k)
    if n == 0:
        raise ValueError("pack_conv1d_h3_weight: C_in must be a multiple of 32")
    out = torch.empty(n, dtype=torch.uint8)
    _lib.check(L.itts_pack_conv1d_h3_weight(_lib.ptr(w), Cout, Cin, k, _lib.ptr(out)), "itts_pack_conv1d_h3_weight")
    return out


def conv1d_h3(x, w3_packed, bias, Cout, k, dilation=1, res=None, lens=None, len_mult=1, out=None, acc_mode=0, div=1.0):
    """the f16 x 3 split-operand Conv1d (`same` zero padding) as a unit op; w3_packed = pack_conv1d_h3_weight(w) on x's device"""
    x = x.float().contiguous()
    B, Cin, T = x.shape
    y = out if out is not None else torch.empty(B, Cout, T, dtype=torch.float32, device=x.device)
    lens_t = None if lens is None else torch.as_tensor(lens, dtype=torch.int32, device=x.device).contiguous()
    L = _lib.lib()
    scratch = torch.empty(L.itts_conv1d_h3_scratch_bytes(B, Cin, T), dtype=torch.uint8, device=x.device)
    with _lib.on_device(x.device):
        _lib.check(L.itts_conv1d_h3_forward(_lib.ptr(x), _lib.ptr(w3_packed), _lib.ptr(bias), _lib.ptr(res), _lib.ptr(y), B, Cin, Cout, T, k,
                                            dilation, _lib.ptr(lens_t), len_mult, acc_mode, float(div), _lib.ptr(scratch),
                                            _lib.stream_ptr(x.device)), "itts_conv1d_h3_forward")
        return y


def pack_conv1d_x3_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout][Cin][k] f32 -> the three bf16 plane fragment streams of the bf16 x 3 conv (uint8 host tensor)"""
    w = w.detach().to("cpu", torch.float32).contiguous()
    Cout, Cin, k = w.shape
    L = _lib.lib()
    n = L.itts_conv1d_x3_packed_bytes(Cout, Cin, k)
    if n == 0:
        raise ValueError("pack_conv1d_x3_weight: C_in must be a multiple of 32")
    out = torch.empty(n, dtype=torch.uint8)
    _lib.check(L.itts_pack_conv1d_x3_weight(_lib.ptr(w), Cout, Cin, k, _lib.ptr(out)), "itts_pack_conv1d_x3_weight")
    return out


def conv1d_x3(x, w3_packed, bias, Cout, k, dilation=1, res=None, lens=None, len_mult=1, out=None, acc_mode=0, div=1.0):
    """the bf16 x 3 plane-operand Conv1d (`same` zero padding) as a unit op; w3_packed = pack_conv1d_x3_weight(w) on x's device"""
    x = x.float().contiguous()
    B, Cin, T = x.shape
    y = out if out is not None else torch.empty(B, Cout, T, dtype=torch.float32, device=x.device)
    lens_t = None if lens is None else torch.as_tensor(lens, dtype=torch.int32, device=x.device).contiguous()
    L = _lib.lib()
    scratch = torch.empty(L.itts_conv1d_x3_scratch_bytes(B, Cin, T), dtype=torch.uint8, device=x.device)
    with _lib.on_device(x.device):
        _lib.check(L.itts_conv1d_x3_forward(_lib.ptr(x), _lib.ptr(w3_packed), _lib.ptr(bias), _lib.ptr(res), _lib.ptr(y), B, Cin, Cout, T, k,
                                            dilation, _lib.ptr(lens_t), len_mult, acc_mode, float(div), _lib.ptr(scratch),
                                            _lib.stream_ptr(x.device)), "itts_conv1d_x3_forward")
        return y


def pack_convT_weight(w: torch.Tensor, u: int) -> torch.Tensor:
    w = w.detach().to("cpu", torch.float32).contiguous()
    Cin, Cout, k = w.shape
    L = _lib.lib()
    per = L.itts_packed_conv_floats(Cout, Cin, k // u)      # k/u taps per phase
    out = torch.empty(per * u, dtype=torch.float32)
    for r in range(u):
        _lib.check(L.itts_pack_convT_weight(_lib.ptr(w), Cin, Cout, k, u, r, C.c_void_p(out.data_ptr() + 4 * per * r)),
                   "itts_pack_convT_weight")
    return out


def conv1d(x, w_packed, bias, Cout, k, dilation=1, res=None, lens=None, len_mult=1, out=None, acc_mode=0, div=1.0):
    x = x.float().contiguous()
    B, Cin, T = x.shape
    y = out if out is not None else torch.empty(B, Cout, T, dtype=torch.float32, device=x.device)
    lens_t = None if lens is None else torch.as_tensor(lens, dtype=torch.int32, device=x.device).contiguous()
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().itts_conv1d_forward(_lib.ptr(x), _lib.ptr(w_packed), _lib.ptr(bias), None, _lib.ptr(res),
                                                  _lib.ptr(y), B, Cin, Cout, T, k, dilation, _lib.ptr(lens_t), len_mult,
                                                  acc_mode, float(div), _lib.stream_ptr(x.device)), "itts_conv1d_forward")
        return y


def conv_transpose1d(x, w_packed, bias, Cout, k, u, lens=None):
    x = x.float().contiguous()
    B, Cin, T = x.shape
    y = torch.zeros(B, Cout, T * u, dtype=torch.float32, device=x.device)
    lens_t = None if lens is None else torch.as_tensor(lens, dtype=torch.int32, device=x.device).contiguous()
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().itts_conv_transpose1d_forward(_lib.ptr(x), _lib.ptr(w_packed), _lib.ptr(bias), None,
                                                            _lib.ptr(y), B, Cin, Cout, T, k, u, _lib.ptr(lens_t), 1,
                                                            _lib.stream_ptr(x.device)), "itts_conv_transpose1d_forward")
        return y
