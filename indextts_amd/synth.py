"""Random-init weights with the reference checkpoint names and shapes, for benchmarking without checkpoints.

There are no model checkpoints in the build/bench environment (SURVEY.md header), so `bench.py` times the engine on
seeded random weights of the IndexTTS-2.5 architecture.  This is NOT the test oracle's generator (tests use
oracle.*.synth_weights); it lives in the package so that the product/bench path never imports `oracle/`.
"""
import math
from typing import Dict

import torch

GPT_V25 = dict(layers=24, model_dim=1280, heads=20, max_text_tokens=600, max_mel_tokens=1815, number_text_tokens=12000,
               number_mel_codes=8194, start_mel_token=8192, stop_mel_token=8193, start_text_token=0, stop_text_token=1,
               max_conditioning_inputs=1, types=1)
N_LANGS = 101

BIGVGAN_V2_22K = dict(num_mels=80, upsample_rates=[4, 4, 2, 2, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4],
                      upsample_initial_channel=1536, resblock_kernel_sizes=[3, 7, 11],
                      resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], activation="snakebeta",
                      snake_logscale=True, use_tanh_at_final=False, use_bias_at_final=False, resblock="1",
                      sampling_rate=22050, hop_size=256)


def gpt_weights(cfg: dict = GPT_V25, seed: int = 1234, suppress_eos: bool = False) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    D, L, V = cfg["model_dim"], cfg["layers"], cfg["number_mel_codes"]
    sd = {}
    rn = lambda *s, std=0.02: torch.randn(*s, generator=g) * std
    for i in range(L):
        p = f"gpt.h.{i}."
        for ln in ("ln_1", "ln_2"):
            sd[p + ln + ".weight"] = 1.0 + rn(D, std=0.05)
            sd[p + ln + ".bias"] = rn(D)
        sd[p + "attn.c_attn.weight"] = rn(D, 3 * D, std=0.05)
        sd[p + "attn.c_attn.bias"] = rn(3 * D)
        sd[p + "attn.c_proj.weight"] = rn(D, D, std=0.05 / math.sqrt(2 * L))
        sd[p + "attn.c_proj.bias"] = rn(D)
        sd[p + "mlp.c_fc.weight"] = rn(D, 4 * D, std=0.05)
        sd[p + "mlp.c_fc.bias"] = rn(4 * D)
        sd[p + "mlp.c_proj.weight"] = rn(4 * D, D, std=0.05 / math.sqrt(2 * L))
        sd[p + "mlp.c_proj.bias"] = rn(D)
    for ln in ("gpt.ln_f", "final_norm"):
        sd[ln + ".weight"] = 1.0 + rn(D, std=0.05)
        sd[ln + ".bias"] = rn(D)
    sd["mel_head.weight"] = rn(V, D, std=0.08)
    sd["mel_head.bias"] = rn(V)
    if suppress_eos:
        sd["mel_head.bias"][cfg["stop_mel_token"]] = -1e4     # fixed-length decode for timing
    sd["mel_embedding.weight"] = rn(V, D, std=0.5)
    sd["mel_pos_embedding.emb.weight"] = rn(cfg["max_mel_tokens"] + 2 + cfg["max_conditioning_inputs"], D, std=0.3)
    sd["text_embedding.weight"] = rn(cfg["number_text_tokens"] * cfg["types"] + 1, D, std=0.5)
    sd["text_pos_embedding.emb.weight"] = rn(cfg["max_text_tokens"] + 2, D, std=0.3)
    sd["lang_embedding.weight"] = rn(N_LANGS, D, std=0.3)
    sd["spk_emb_proj.weight"] = rn(D, 192, std=0.05)
    sd["spk_emb_proj.bias"] = rn(D)
    return sd


def kaiser_sinc_filter12() -> torch.Tensor:
    """12-tap Kaiser-windowed sinc, cutoff 0.25, half-width 0.3 (the buffer UpSample1d/DownSample1d register)."""
    half, K, cutoff, hw = 6, 12, 0.25, 0.3
    A = 2.285 * (half - 1) * math.pi * 4 * hw + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21) if A >= 21 else 0.0)
    win = torch.kaiser_window(K, beta=beta, periodic=False)
    t = torch.arange(-half, half) + 0.5
    f = 2 * cutoff * win * torch.sinc(2 * cutoff * t)
    return (f / f.sum()).float()


def bigvgan_weights(h: dict = BIGVGAN_V2_22K, seed: int = 1234, post_gain: float = 0.04) -> Dict[str, torch.Tensor]:
    """Variance-preserving init (std 1/sqrt(C_in*k)) so the waveform has RMS ~0.2 instead of collapsing to zero."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    filt = kaiser_sinc_filter12().view(1, 1, 12)

    def conv(name, cout, cin, k, bias=True, gain=1.0):
        sd[name + ".weight"] = torch.randn(cout, cin, k, generator=g) * (gain / math.sqrt(cin * k))
        if bias:
            sd[name + ".bias"] = torch.randn(cout, generator=g) * 0.02

    def act(name, ch):
        sd[name + ".act.alpha"] = torch.rand(ch, generator=g) - 0.5
        sd[name + ".act.beta"] = torch.rand(ch, generator=g) - 0.5
        sd[name + ".upsample.filter"] = filt.clone()
        sd[name + ".downsample.lowpass.filter"] = filt.clone()

    c0 = h["upsample_initial_channel"]
    conv("conv_pre", c0, h["num_mels"], 7)
    nk = len(h["resblock_kernel_sizes"])
    ch = c0
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        cin, ch = c0 // 2 ** i, c0 // 2 ** (i + 1)
        sd[f"ups.{i}.0.weight"] = torch.randn(cin, ch, k, generator=g) / math.sqrt(cin * k / u)
        sd[f"ups.{i}.0.bias"] = torch.randn(ch, generator=g) * 0.02
        for j, kk in enumerate(h["resblock_kernel_sizes"]):
            n = i * nk + j
            nd = len(h["resblock_dilation_sizes"][j])
            for d in range(nd):
                conv(f"resblocks.{n}.convs1.{d}", ch, ch, kk, gain=0.7)
                conv(f"resblocks.{n}.convs2.{d}", ch, ch, kk, gain=0.5)
            for m in range(2 * nd):
                act(f"resblocks.{n}.activations.{m}", ch)
    act("activation_post", ch)
    conv("conv_post", 1, ch, 7, bias=h.get("use_bias_at_final", True), gain=post_gain)
    return sd
