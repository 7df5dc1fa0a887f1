"""ctypes binding of the C ABI declared in include/indextts_hip.h.

The product path has NO fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libindextts_hip.so")

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
vp = C.c_void_p


class BigVGANConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("upsample_initial_channel", C.c_int32), ("num_upsamples", C.c_int32),
        ("upsample_rates", C.c_int32 * 8), ("upsample_kernel_sizes", C.c_int32 * 8), ("num_kernels", C.c_int32),
        ("resblock_kernel_sizes", C.c_int32 * 4), ("num_dilations", C.c_int32),
        ("resblock_dilations", (C.c_int32 * 4) * 4), ("snake_logscale", C.c_int32), ("activation", C.c_int32),
        ("use_tanh_at_final", C.c_int32), ("use_bias_at_final", C.c_int32), ("cond_dim", C.c_int32),
        ("cond_in_each_up_layer", C.c_int32),
    ]


class S2MelConfig(C.Structure):
    _fields_ = [("hidden_dim", C.c_int32), ("num_heads", C.c_int32), ("depth", C.c_int32), ("in_channels", C.c_int32),
                ("wavenet_hidden", C.c_int32), ("wavenet_layers", C.c_int32), ("wavenet_kernel", C.c_int32),
                ("wavenet_dilation_rate", C.c_int32), ("precision", C.c_int32), ("norm_eps", C.c_float)]


class FbankConfig(C.Structure):
    _fields_ = [("frame_length", C.c_int32), ("hop", C.c_int32), ("n_fft", C.c_int32), ("n_mels", C.c_int32), ("pad", C.c_int32),
                ("remove_dc", C.c_int32), ("power", C.c_int32), ("take_log", C.c_int32), ("layout", C.c_int32),
                ("preemphasis", C.c_float), ("mag_eps", C.c_float), ("floor", C.c_float), ("scale", C.c_float)]


class GPTConfig(C.Structure):
    _fields_ = [("layers", C.c_int32), ("model_dim", C.c_int32), ("heads", C.c_int32), ("vocab", C.c_int32),
                ("n_mel_pos", C.c_int32), ("precision", C.c_int32), ("start_mel_token", C.c_int32),
                ("stop_mel_token", C.c_int32), ("ln_eps", C.c_float)]


ABI_VERSION = 12         # include/indextts_hip.h ITTS_ABI_VERSION


class GenParams(C.Structure):
    _fields_ = [("do_sample", C.c_int32), ("num_beams", C.c_int32), ("top_k", C.c_int32),
                ("min_tokens_to_keep", C.c_int32), ("max_new_tokens", C.c_int32), ("pos_offset", C.c_int32),
                ("top_p", C.c_float), ("temperature", C.c_float), ("repetition_penalty", C.c_float),
                ("length_penalty", C.c_float), ("typical_mass", C.c_float), ("reserved", C.c_int32),
                ("seed", C.c_uint64)]


# name -> (restype, argtypes); every symbol include/indextts_hip.h declares must appear here
SIGNATURES = {
    "itts_abi_version": (C.c_int, []),
    "itts_last_error": (C.c_char_p, []),
    "itts_device_count": (C.c_int, []),
    "itts_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "itts_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "itts_reset_options": (C.c_int, []),
    "itts_option_count": (C.c_int, []),
    "itts_option_name": (C.c_char_p, [C.c_int]),
    "itts_option_doc": (C.c_char_p, [C.c_int]),
    "itts_option_default": (C.c_int, [C.c_int]),
    "itts_aa_act_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]),
    "itts_packed_conv_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "itts_pack_conv1d_weight": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
    "itts_pack_convT_weight": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "itts_conv1d_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      vp, C.c_int, C.c_int, C.c_float, vp]),
    "itts_conv_transpose1d_forward": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_int, vp, C.c_int, vp]),
    "itts_bigvgan_create": (C.c_int, [C.POINTER(BigVGANConfig), C.POINTER(vp)]),
    "itts_bigvgan_device": (C.c_int, [vp]),
    "itts_bigvgan_load_tensor": (C.c_int, [vp, C.c_char_p, vp, c_i64p, C.c_int]),
    "itts_bigvgan_set_conv_mode": (C.c_int, [vp, C.c_int, C.c_int]),
    "itts_bigvgan_range_check": (C.c_int, [vp]),
    "itts_conv1d_h3_packed_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "itts_pack_conv1d_h3_weight": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
    "itts_conv1d_h3_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "itts_conv1d_h3_forward": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int,
                                         C.c_float, vp, vp]),
    "itts_conv1d_x3_packed_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "itts_pack_conv1d_x3_weight": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
    "itts_conv1d_x3_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "itts_conv1d_x3_forward": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int,
                                         C.c_float, vp, vp]),
    "itts_bigvgan_finalize": (C.c_int, [vp]),
    "itts_bigvgan_destroy": (None, [vp]),
    "itts_bigvgan_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int]),
    "itts_bigvgan_forward": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "itts_bigvgan_stream_open": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "itts_bigvgan_stream_workspace_bytes": (C.c_size_t, [vp]),
    "itts_bigvgan_stream_push": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.POINTER(C.c_int32), vp, C.c_size_t, vp]),
    "itts_bigvgan_stream_close": (None, [vp]),
    "itts_bigvgan_set_profiling": (C.c_int, [vp, C.c_int]),
    "itts_bigvgan_profile_read": (C.c_int, [vp, vp, vp, vp, vp]),
    "itts_bigvgan_profile_records": (C.c_int, [vp, vp, C.c_int]),
    "itts_packed_gemm_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "itts_pack_gemm_weight": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "itts_gpt_create": (C.c_int, [C.POINTER(GPTConfig), C.POINTER(vp)]),
    "itts_gpt_device": (C.c_int, [vp]),
    "itts_gpt_load_tensor": (C.c_int, [vp, C.c_char_p, vp, c_i64p, C.c_int]),
    "itts_gpt_finalize": (C.c_int, [vp]),
    "itts_gpt_destroy": (None, [vp]),
    "itts_gpt_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int, C.c_int]),
    "itts_gpt_generate": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.POINTER(GenParams), c_i32p, C.c_int, vp, vp,
                                    C.POINTER(C.c_int32), vp, C.c_size_t, C.c_int, vp]),
    "itts_gpt_generate_chunk": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.POINTER(GenParams), c_i32p, C.c_int, vp, vp, C.c_int32,
                                          C.POINTER(C.c_int32), vp, C.c_size_t, C.c_int, vp]),
    "itts_gpt_beam_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "itts_gpt_generate_beam": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(GenParams), c_i32p, C.c_int, vp,
                                         vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int32), vp, C.c_size_t, C.c_int, vp]),
    "itts_gpt_admit_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int]),
    "itts_gpt_admit_rows": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_size_t, vp, C.c_size_t, vp]),
    "itts_gpt_last_timing": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "itts_gpt_graph_stats": (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "itts_gpt_set_compaction": (C.c_int, [vp, C.c_int, C.c_int]),
    "itts_gpt_set_row_limits": (C.c_int, [vp, vp, C.c_int]),
    "itts_gpt_set_chunk_return": (C.c_int, [vp, C.c_int]),
    "itts_gpt_compaction_stats": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "itts_gpt_forward_latent": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, C.c_size_t, vp]),
    "itts_gemm_tile_occupancy": (C.c_int, [C.c_int, C.POINTER(C.c_int32)]),
    "itts_s2mel_create": (C.c_int, [C.POINTER(S2MelConfig), C.POINTER(vp)]),
    "itts_s2mel_device": (C.c_int, [vp]),
    "itts_s2mel_load_tensor": (C.c_int, [vp, C.c_char_p, vp, c_i64p, C.c_int]),
    "itts_s2mel_set_tail": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int]),
    "itts_s2mel_finalize": (C.c_int, [vp]),
    "itts_s2mel_destroy": (None, [vp]),
    "itts_s2mel_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int, C.c_int]),
    "itts_s2mel_mods_per_step": (C.c_int, [vp]),
    "itts_s2mel_estimator": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_size_t, vp]),
    "itts_s2mel_solve": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(C.c_float), C.c_float, vp, C.c_size_t, vp]),
    "itts_s2mel_set_profiling": (C.c_int, [vp, C.c_int]),
    "itts_s2mel_profile_read": (C.c_int, [vp, vp, vp, vp]),
    "itts_s2mel_set_trace": (C.c_int, [vp, vp, C.c_int]),
    "itts_s2mel_trace_count": (C.c_int, [vp]),
    "itts_s2mel_trace_wanted": (C.c_int, [vp]),
    "itts_s2mel_trace_label": (C.c_char_p, [vp, C.c_int]),
    "itts_s2mel_set_capture": (C.c_int, [vp, vp, C.c_size_t, C.c_char_p]),
    "itts_s2mel_capture_offset": (C.c_longlong, [vp, C.c_int, C.POINTER(C.c_size_t)]),
    "itts_s2mel_attention_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "itts_s2mel_attention_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp,
                                               C.c_size_t, vp]),
    "itts_vq_project_forward": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "itts_vq_search_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "itts_tok_gather_conv_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "itts_tok_dwconv_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "itts_tok_gelu_forward": (C.c_int, [vp, C.c_size_t, vp]),
    "itts_attention_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "itts_attention_relkey_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "itts_tok_dwconv_causal_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "itts_tok_glu_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "itts_tok_act_forward": (C.c_int, [vp, C.c_size_t, C.c_int, vp]),
    "itts_tok_l2norm_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    "itts_tok_affine_forward": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "itts_tok_ctxpool_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "itts_tok_gate_forward": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "itts_tok_attnstats_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    "itts_tok_statspool_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "itts_tok_scale_residual_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "itts_tok_groupnorm_mish_forward": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    "itts_gemm_forward": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "itts_layernorm_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    "itts_gemm_ln_forward": (C.c_int, [vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "itts_fbank_frames": (C.c_int, [C.POINTER(FbankConfig), C.c_int]),
    "itts_fbank_forward": (C.c_int, [vp, C.c_int, C.c_int, C.c_int64, C.POINTER(FbankConfig), vp, vp, vp, vp, C.c_int, C.c_int64, vp]),
    "itts_resample_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, vp]),
    "itts_tok_colnorm_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
}

_lib = None


class HipEngineError(RuntimeError):
    pass


def lib():
    """Load libindextts_hip.so (fails loudly; there is no CPU fallback on the product path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipEngineError(
                f"{LIB_PATH} is missing: build it with `python indextts_amd/build.py` (hipcc, gfx950). "
                "The engine has no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64.  The engine must share THAT runtime instance (device
        # pointers, streams and events cross the boundary): importing torch first makes the loader bind the library's
        # libamdhip64.so.7 dependency to the copy torch already mapped.  Loaded the other way round the process ends up with
        # two HSA runtimes and the second one sees no device.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.itts_abi_version() != ABI_VERSION:
            raise HipEngineError("libindextts_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().itts_last_error().decode(errors="replace")
        raise HipEngineError(f"{what} failed (code {rc}): {msg}")


def set_option(name: str, value: int):
    """Process-wide engine option (include/indextts_hip.h: the table above itts_set_option)."""
    check(lib().itts_set_option(name.encode(), int(value)), f"set_option({name})")


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(lib().itts_get_option(name.encode(), C.byref(v)), f"get_option({name})")
    return int(v.value)


def reset_options():
    check(lib().itts_reset_options(), "reset_options")


def options() -> dict:
    """name -> (current value, default, doc) of every engine option."""
    L = lib()
    out = {}
    for i in range(L.itts_option_count()):
        name = L.itts_option_name(i).decode()
        out[name] = (get_option(name), int(L.itts_option_default(i)), L.itts_option_doc(i).decode())
    return out


class option_scope:
    """with option_scope(decode_fuse_ln=0): ...  -- sets options for the block and restores the previous values."""

    def __init__(self, **kv):
        self.kv = kv
        self.prev = {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.prev[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_option(k, v)
        return False


def ptr(t):
    """Raw address of a torch tensor (device or host) or None."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """torch's current stream ON `device` (default: the current device) as a hipStream_t."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_device(device):
    """Context manager making `device` the current HIP device (handles are bound to the device current at *_create)."""
    import contextlib
    import torch
    d = torch.device(device) if device is not None else None
    if d is None or d.type != "cuda" or not torch.cuda.is_available():
        return contextlib.nullcontext()
    return torch.cuda.device(d)
