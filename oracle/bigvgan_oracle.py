"""CPU oracle for the BigVGAN vocoder hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a plain fp32 CPU restatement of the reference vocoder arithmetic.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it; the product path (`indextts_amd/`) never does.

Pinning: `tests/golden/bigvgan_*.npz` were produced by importing the reference
classes themselves (`tools/make_golden_bigvgan.py`, run in the build container
where /root/reference exists) and `tests/test_oracle_bigvgan.py` checks this
restatement against them -> parity is PINNED for the vocoder.

Reference sources restated (paths relative to the reference repo root):
  * BigVGAN-v2 generator forward      indextts/s2mel/modules/bigvgan/bigvgan.py:360-386
  * AMPBlock1.forward                 indextts/s2mel/modules/bigvgan/bigvgan.py:132-141
  * Activation1d (up -> act -> down)  .../alias_free_activation/torch/act.py:25-30
  * UpSample1d / DownSample1d         .../alias_free_activation/torch/resample.py:29-38,55-58
  * LowPassFilter1d / kaiser filter   .../alias_free_activation/torch/filter.py:30-62,93-101
  * SnakeBeta / Snake                 indextts/s2mel/modules/bigvgan/activations.py:49-53,107-117
  * fused CUDA activation (same math) .../alias_free_activation/cuda/anti_alias_activation_cuda.cu:43-179
  * v1 generator (speaker cond, tanh) indextts/BigVGAN/models.py:201-250

The anti-aliased activation is written out as explicit index arithmetic (no
conv_transpose call) so that it is an independent statement of the same math
the HIP kernel implements:

    u[2q]   = 2 * sum_{j=0..5} fu[1+2j]  * x[clamp(q+2-j)]        (even phase)
    u[2q+1] = 2 * sum_{j=0..5} fu[2j]    * x[clamp(q+3-j)]        (odd phase)
    v[i]    = u[i] + 1/(exp(beta)+1e-9) * sin(u[i]*exp(alpha))^2
    y[t]    = sum_{j=0..11} fd[j] * v[clamp(2t + j - 5, 0, 2T-1)]

(clamp on x is to [0, T-1]: replicate padding 5/5; clamp on v: replicate 5/6).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

NO_DIV_BY_ZERO = 1e-9


# ----------------------------------------------------------------------------
# filters  (filter.py:30-62)
# ----------------------------------------------------------------------------
def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> torch.Tensor:
    """Kaiser-windowed sinc low-pass, normalised to unit DC gain. Returns (kernel_size,)."""
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    if even:
        time = torch.arange(-half_size, half_size) + 0.5
    else:
        time = torch.arange(kernel_size) - half_size
    if cutoff == 0:
        return torch.zeros(kernel_size)
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    filt = filt / filt.sum()
    return filt.to(torch.float32)


def default_filter() -> torch.Tensor:
    """The 12-tap filter both UpSample1d(2) and DownSample1d(2) build (resample.py:23-26,49-54)."""
    return kaiser_sinc_filter1d(cutoff=0.25, half_width=0.3, kernel_size=12)


# ----------------------------------------------------------------------------
# anti-aliased activation, explicit index form
# ----------------------------------------------------------------------------
def upsample2x(x: torch.Tensor, fu: torch.Tensor) -> torch.Tensor:
    """x (B,C,T) -> (B,C,2T); replicate pad 5, 12-tap polyphase, gain 2."""
    B, C, T = x.shape
    q = torch.arange(T)
    even = torch.zeros_like(x)
    odd = torch.zeros_like(x)
    for j in range(6):
        idx_e = (q + 2 - j).clamp(0, T - 1)
        idx_o = (q + 3 - j).clamp(0, T - 1)
        even = even + fu[1 + 2 * j] * x[..., idx_e]
        odd = odd + fu[2 * j] * x[..., idx_o]
    u = torch.stack([even, odd], dim=-1).reshape(B, C, 2 * T)
    return 2.0 * u


def snake_beta(u: torch.Tensor, alpha_log: torch.Tensor, beta_log: torch.Tensor,
               logscale: bool = True) -> torch.Tensor:
    a = alpha_log.view(1, -1, 1)
    b = beta_log.view(1, -1, 1)
    if logscale:
        a = torch.exp(a)
        b = torch.exp(b)
    return u + (1.0 / (b + NO_DIV_BY_ZERO)) * torch.sin(u * a) ** 2


def downsample2x(v: torch.Tensor, fd: torch.Tensor) -> torch.Tensor:
    """v (B,C,2T) -> (B,C,T); replicate pad 5 left / 6 right, 12-tap, stride 2."""
    T2 = v.shape[-1]
    T = T2 // 2
    t = torch.arange(T)
    y = torch.zeros(v.shape[0], v.shape[1], T, dtype=v.dtype)
    for j in range(12):
        idx = (2 * t + j - 5).clamp(0, T2 - 1)
        y = y + fd[j] * v[..., idx]
    return y


# bench.py's cpu_baseline leg times the restatement as the reference's CPU path: with TIMING_MODE on, the two resamplers run as strided depthwise
# convolutions (the form the reference's torch modules use, resample.py:29-38,55-58) instead of the index arithmetic above, which is 2.7 x slower on
# CPU (profiles/r03z_cpu/reference_cpu_timing.log).  Same taps, same padding; summation order differs (tests/test_oracle_timing_mode.py bounds the
# difference).  The CHECKER (tests, smoke) always runs the index form.
TIMING_MODE = False


def _resample_conv(x: torch.Tensor, fu: torch.Tensor, fd: torch.Tensor, act) -> torch.Tensor:
    C = x.shape[1]
    u = 2.0 * F.conv_transpose1d(F.pad(x, (5, 5), mode="replicate"), fu.view(1, 1, 12).expand(C, 1, 12), stride=2, groups=C)[..., 15:-15]
    return F.conv1d(F.pad(act(u), (5, 6), mode="replicate"), fd.view(1, 1, 12).expand(C, 1, 12), stride=2, groups=C)


def activation1d(x: torch.Tensor, alpha_log: torch.Tensor, beta_log: torch.Tensor,
                 fu: Optional[torch.Tensor] = None, fd: Optional[torch.Tensor] = None,
                 logscale: bool = True) -> torch.Tensor:
    fu = default_filter() if fu is None else fu.reshape(-1).float()
    fd = default_filter() if fd is None else fd.reshape(-1).float()
    if TIMING_MODE:
        return _resample_conv(x, fu, fd, lambda u: snake_beta(u, alpha_log, beta_log, logscale))
    return downsample2x(snake_beta(upsample2x(x, fu), alpha_log, beta_log, logscale), fd)


# ----------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------
V2_HPARAMS = dict(
    num_mels=80,
    upsample_rates=[4, 4, 2, 2, 2, 2],
    upsample_kernel_sizes=[8, 8, 4, 4, 4, 4],
    upsample_initial_channel=1536,
    resblock_kernel_sizes=[3, 7, 11],
    resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    activation="snakebeta",
    snake_logscale=True,
    use_tanh_at_final=False,
    use_bias_at_final=False,
    sampling_rate=22050,
)


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept checkpoints saved before `remove_weight_norm()` (bigvgan.py:388-400).

    weight = g * v / ||v||  with the norm over every dim but 0 (torch weight_norm, dim=0),
    for both the legacy `weight_g/weight_v` and the parametrizations layout.
    """
    out = dict(sd)
    for k in list(sd.keys()):
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            g, v = sd[k], sd[base + ".weight_v"]
        elif k.endswith(".parametrizations.weight.original0"):
            base = k[: -len(".parametrizations.weight.original0")]
            g, v = sd[k], sd[base + ".parametrizations.weight.original1"]
        else:
            continue
        dims = tuple(range(1, v.dim()))
        w = v * (g / v.norm(2, dim=dims, keepdim=True))
        out[base + ".weight"] = w
        for suf in (".weight_g", ".weight_v", ".parametrizations.weight.original0",
                    ".parametrizations.weight.original1"):
            out.pop(base + suf, None)
    return out


def synth_weights(h: dict, seed: int = 1234, cond_dim: int = 0, in_dim: Optional[int] = None,
                  post_gain: float = 0.04) -> Dict[str, torch.Tensor]:
    """Seeded synthetic generator weights with the reference state-dict names.

    Variance-preserving init (std = 1/sqrt(C_in*k)) instead of the reference's
    N(0, 0.01) `init_weights` (utils.py:45-48): through ~110 layers N(0,0.01)
    collapses the signal to ~0 and would make an absolute-RMS gate vacuous
    (SURVEY.md section 8d).  alpha/beta ~ U(-0.5, 0.5) in log scale.
    `cond_dim > 0` adds the v1 speaker-conditioning 1x1 convs (models.py:191-197).
    `post_gain` 0.04 gives a waveform RMS of ~0.15-0.2 that almost never touches the final
    clamp; 0.35 gives a loud signal where ~50 % of samples clamp (golden case "loud").
    """
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, k, bias=True, gain=1.0):
        std = gain / math.sqrt(cin * k)
        sd[name + ".weight"] = torch.randn(cout, cin, k, generator=g) * std
        if bias:
            sd[name + ".bias"] = torch.randn(cout, generator=g) * 0.02

    def act(name, ch):
        sd[name + ".act.alpha"] = torch.rand(ch, generator=g) - 0.5
        sd[name + ".act.beta"] = torch.rand(ch, generator=g) - 0.5
        sd[name + ".upsample.filter"] = default_filter().view(1, 1, 12)
        sd[name + ".downsample.lowpass.filter"] = default_filter().view(1, 1, 12)

    c0 = h["upsample_initial_channel"]
    conv("conv_pre", c0, in_dim if in_dim is not None else h["num_mels"], 7)
    if cond_dim:
        conv("cond_layer", c0, cond_dim, 1, gain=0.3)
    ch = c0
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        # ConvTranspose1d weight is (C_in, C_out, k); each output sample sums C_in*k/u taps
        std = 1.0 / math.sqrt(cin * k / u)
        sd[f"ups.{i}.0.weight"] = torch.randn(cin, cout, k, generator=g) * std
        sd[f"ups.{i}.0.bias"] = torch.randn(cout, generator=g) * 0.02
        if cond_dim:
            conv(f"conds.{i}", cout, cond_dim, 1, gain=0.3)
        ch = cout
        for j, kk in enumerate(h["resblock_kernel_sizes"]):
            n = i * nk + j
            for d in range(len(h["resblock_dilation_sizes"][j])):
                # residual branches: damp so the residual stream stays O(1)
                conv(f"resblocks.{n}.convs1.{d}", ch, ch, kk, gain=0.7)
                conv(f"resblocks.{n}.convs2.{d}", ch, ch, kk, gain=0.5)
            for m in range(2 * len(h["resblock_dilation_sizes"][j])):
                act(f"resblocks.{n}.activations.{m}", ch)
    act("activation_post", ch)
    conv("conv_post", 1, ch, 7, bias=h.get("use_bias_at_final", True), gain=post_gain)
    return sd


# ----------------------------------------------------------------------------
# generator forward
# ----------------------------------------------------------------------------
def _act(sd, name, x, logscale):
    return activation1d(x, sd[name + ".act.alpha"], sd[name + ".act.beta"],
                        sd.get(name + ".upsample.filter"), sd.get(name + ".downsample.lowpass.filter"),
                        logscale)


def amp_block1(sd, n: int, x: torch.Tensor, k: int, dilations: Sequence[int], logscale: bool) -> torch.Tensor:
    """bigvgan.py:132-141 : x = x + conv2(act2(conv1_d(act1(x)))) for each dilation."""
    for di, d in enumerate(dilations):
        xt = _act(sd, f"resblocks.{n}.activations.{2 * di}", x, logscale)
        xt = F.conv1d(xt, sd[f"resblocks.{n}.convs1.{di}.weight"], sd.get(f"resblocks.{n}.convs1.{di}.bias"),
                      dilation=d, padding=(k * d - d) // 2)
        xt = _act(sd, f"resblocks.{n}.activations.{2 * di + 1}", xt, logscale)
        xt = F.conv1d(xt, sd[f"resblocks.{n}.convs2.{di}.weight"], sd.get(f"resblocks.{n}.convs2.{di}.bias"),
                      padding=(k - 1) // 2)
        x = xt + x
    return x


def bigvgan_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, h: dict = V2_HPARAMS,
                    spk: Optional[torch.Tensor] = None, taps: Optional[List] = None) -> torch.Tensor:
    """Generator forward on CPU fp32.

    x: (B, C_in, T) mel (v2) or GPT latent already transposed (v1).
    spk: (B, cond_dim, 1) speaker embedding for the v1 variant (models.py:216-231), else None.
    taps: if a list is passed, the stage outputs are appended (debugging aid for kernel bring-up).
    """
    sd = fold_weight_norm(sd)
    x = x.float()
    logscale = h.get("snake_logscale", True)
    nk = len(h["resblock_kernel_sizes"])
    x = F.conv1d(x, sd["conv_pre.weight"], sd.get("conv_pre.bias"), padding=3)
    if spk is not None:
        x = x + F.conv1d(spk, sd["cond_layer.weight"], sd.get("cond_layer.bias"))
    if taps is not None:
        taps.append(x)
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.conv_transpose1d(x, sd[f"ups.{i}.0.weight"], sd.get(f"ups.{i}.0.bias"),
                               stride=u, padding=(k - u) // 2)
        if spk is not None and f"conds.{i}.weight" in sd:
            x = x + F.conv1d(spk, sd[f"conds.{i}.weight"], sd.get(f"conds.{i}.bias"))
        xs = None
        for j, kk in enumerate(h["resblock_kernel_sizes"]):
            r = amp_block1(sd, i * nk + j, x, kk, h["resblock_dilation_sizes"][j], logscale)
            xs = r if xs is None else xs + r        # (k=3)+(k=7)+(k=11) order, then /3 (bigvgan.py:369-375)
        x = xs / nk
        if taps is not None:
            taps.append(x)
    x = _act(sd, "activation_post", x, logscale)
    x = F.conv1d(x, sd["conv_post.weight"], sd.get("conv_post.bias"), padding=3)
    if h.get("use_tanh_at_final", True):
        x = torch.tanh(x)
    else:
        x = torch.clamp(x, min=-1.0, max=1.0)
    return x
