"""CPU study (no GPU): what operand precision the flow-matching stage needs for north_star's 1e-4 waveform bar.

The oracle's 25-step CFG Euler solve (oracle/s2mel_oracle.py, production widths: DiT 13 x 512, WaveNet 8 x 512) is run in f32 and with the GEMM /
attention OPERANDS rounded the way a reduced-precision engine mode would round them (accumulation, residual streams, norms, softmax stay f32 --
the contract of the engine's bf16 mode), then both mels go through the BigVGAN oracle; reported: mel RMS error and waveform RMS error vs f32.
  modes: bf16 (the engine's bf16 mode: sanity anchor against the GPU measurement, 7.1e-4 at 517 + 1926 frames), fp16 (11-bit significand on the same
  MFMA rate), and fp16 with chosen GEMMs kept in f32.
usage: s2mel_precision_study.py [prompt_frames] [target_frames] [steps]      (defaults 150 350 25: about a minute per mode on 8 threads)"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import bigvgan_oracle as BO  # noqa: E402
from oracle import s2mel_oracle as S  # noqa: E402

Tp = int(sys.argv[1]) if len(sys.argv) > 1 else 150
Tg = int(sys.argv[2]) if len(sys.argv) > 2 else 350
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 25


class Mode:
    def __init__(self, name, dtype=None, keep_f32=(), only=None):
        self.name, self.dtype, self.keep_f32 = name, dtype, set(keep_f32)
        self.only = None if only is None else set(only)           # when given: ONLY the GEMMs whose name matches are rounded, the rest stay f32

    def rounds(self, name):
        if self.dtype is None:
            return False
        if self.only is not None:
            return any(s in name for s in self.only)
        return not any(s in name for s in self.keep_f32)

    def r(self, x):
        if self.dtype is None:
            return x
        if self.dtype == "bf16x2":                                 # two bf16 planes: 16 significand bits (the first two planes of the x3 split)
            h = x.bfloat16().float()
            return h + (x - h).bfloat16().float()
        return x.to(self.dtype).float()


MODE = Mode("f32")
_TAG = [None]          # which GEMM is running (set by the patched call sites through the weight tensor's identity)
_WNAME = {}
_KEEP = []


class ProxyF:
    """torch.nn.functional with operand rounding on linear / conv1d (big-M calls only: the per-step modulation MLPs act on one row)."""

    def __getattr__(self, k):
        return getattr(TF, k)

    @staticmethod
    def linear(x, w, b=None):
        rows = x.numel() // x.shape[-1]
        name = _WNAME.get(id(w), "?")
        if rows <= 4 or not MODE.rounds(name):
            return TF.linear(x, w, b)
        return TF.linear(MODE.r(x), MODE.r(w), b)

    @staticmethod
    def conv1d(x, w, b=None, **kw):
        name = _WNAME.get(id(w), "conv")
        if x.shape[-1] <= 4 or not MODE.rounds(name):
            return TF.conv1d(x, w, b, **kw)
        return TF.conv1d(MODE.r(x), MODE.r(w), b, **kw)


def attention(sd, prefix, c, x, tab, key_mask):
    B, T, _ = x.shape
    H, hd = c.num_heads, c.head_dim
    q, k, v = S.F.linear(x, sd[prefix + "wqkv.weight"]).split([H * hd] * 3, dim=-1)
    q = S.apply_rope(q.view(B, T, H, hd), tab).transpose(1, 2)
    k = S.apply_rope(k.view(B, T, H, hd), tab).transpose(1, 2)
    v = v.view(B, T, H, hd).transpose(1, 2)
    rr = MODE.r if MODE.rounds("attention_operands") else (lambda t: t)
    s = (rr(q) @ rr(k).transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    y = rr(torch.softmax(s, dim=-1)) @ rr(v)
    return S.F.linear(y.transpose(1, 2).reshape(B, T, H * hd), sd[prefix + "wo.weight"])


def rms(t):
    return float(t.double().pow(2).mean().sqrt())


def main():
    global MODE
    torch.manual_seed(0)
    cfg = S.S2MelConfig()
    sd = S.synth_weights(cfg, 3)
    for k, v in sd.items():
        _WNAME[id(v)] = k
    S.F = ProxyF()
    S.attention = attention
    wn0 = S._wn

    def wn_named(sd_, prefix):                     # weight-normed layers are folded per call: name the folded tensor after its prefix
        w = wn0(sd_, prefix)
        _WNAME[id(w)] = prefix + "weight"
        _KEEP.append(w)                            # keep the tensor alive so that ids are not reused
        return w
    S._wn = wn_named
    g = torch.Generator().manual_seed(11)
    T = Tp + Tg
    x = torch.randn(1, 80, T, generator=g)
    mu = torch.randn(1, T, cfg.content_dim, generator=g)
    prompt = torch.randn(1, 80, Tp, generator=g) * 2 - 4
    style = torch.randn(1, cfg.style_dim, generator=g)
    h = dict(BO.V2_HPARAMS)
    vsd = BO.synth_weights(h, seed=1234)
    modes = [Mode("f32"), Mode("bf16", torch.bfloat16), Mode("fp16", torch.float16),
             Mode("fp16, attention operands f32", torch.float16, ("attention_operands",)),
             Mode("fp16, output head f32 (final_layer.linear, conv2, skip_linear, conv1, res_projection)", torch.float16,
                  ("final_layer.linear", "conv2", "skip_linear", "conv1.weight", "res_projection")),
             Mode("fp16, WaveNet convs f32", torch.float16, ("wavenet",)),
             Mode("bf16, attention operands f32", torch.bfloat16, ("attention_operands",)),
             # candidates for a mixed mode (indices 7..10)
             Mode("only the attention operands (Q, K, V, P) in bf16, every GEMM f32", torch.bfloat16, only=("attention_operands",)),
             Mode("only the attention operands (Q, K, V, P) in fp16, every GEMM f32", torch.float16, only=("attention_operands",)),
             Mode("attention operands + the big GEMMs (wqkv, wo, w1/w3, w2, WaveNet) in fp16; merge / skip / head / final GEMMs f32", torch.float16,
                  only=("attention_operands", "wqkv", "attention.wo", "feed_forward", "in_layers", "res_skip_layers")),
             Mode("attention operands + w1/w3 + w2 in fp16, the rest f32", torch.float16, only=("attention_operands", "feed_forward")),
             # index 11, 12: GEMM operands as TWO bf16 planes (16 bits; 3 plane products hh, hm, mh on the bf16 pipe)
             Mode("every GEMM operand as two bf16 planes (16 bits), attention operands f32", "bf16x2", keep_f32=("attention_operands",)),
             Mode("every GEMM operand as two bf16 planes (16 bits) AND attention operands as two planes", "bf16x2")]
    if os.environ.get("STUDY_SWEEP"):                # one GEMM group at a time in the low precision, everything else f32
        dt_ = torch.float16 if os.environ["STUDY_SWEEP"] == "fp16" else torch.bfloat16
        groups = [("wqkv", ("wqkv",)), ("attention operands (Q, K, V, P)", ("attention_operands",)), ("wo", ("attention.wo",)),
                  ("w1 / w3", ("feed_forward.w1", "feed_forward.w3")), ("w2", ("feed_forward.w2",)), ("skip_in_linear", ("skip_in_linear",)),
                  ("cond_x_merge_linear + cond_projection", ("cond_x_merge_linear", "cond_projection")),
                  ("skip_linear + conv1 + res_projection", ("estimator.skip_linear", "estimator.conv1.", "res_projection")),
                  ("WaveNet in_layers", ("in_layers",)), ("WaveNet res_skip_layers", ("res_skip_layers",)),
                  ("final_layer.linear + conv2", ("final_layer.linear", "estimator.conv2"))]
        modes = [Mode("f32")] + [Mode(f"only {n} in {os.environ['STUDY_SWEEP']}", dt_, only=pat) for n, pat in groups]
    only = os.environ.get("STUDY_MODES")           # e.g. "0,4,5": run the f32 reference and a subset
    if only:
        modes = [modes[int(i)] for i in only.split(",")]
    ref_mel = ref_wav = None
    print(f"oracle CFM 13 x 512 + WaveNet 8 x 512, {Tp} + {Tg} frames, {STEPS} CFG Euler steps; BigVGAN oracle 1536 ch; {torch.get_num_threads()} threads",
          flush=True)
    for m in modes:
        MODE = m
        t0 = time.perf_counter()
        with torch.no_grad():
            mel = S.cfm_solve_euler(sd, cfg, x.clone(), torch.tensor([T]), prompt, mu, style, STEPS, 0.7)[:, :, Tp:]
            MODE = Mode("f32")                      # the vocoder always in f32: only the mel perturbation is carried through it
            wav = BO.bigvgan_forward(vsd, mel, h)
        dt = time.perf_counter() - t0
        if ref_mel is None:
            ref_mel, ref_wav = mel, wav
            print(f"{m.name:90s} mel rms {rms(mel):.3f}  waveform rms {rms(wav):.3f}  ({dt:.0f} s)", flush=True)
            continue
        em, ew = rms(mel - ref_mel), rms(wav - ref_wav)
        print(f"{m.name:90s} mel rms error {em:.3e} (relative {em / rms(ref_mel):.2e})  waveform rms error {ew:.3e}  "
              f"{'UNDER' if ew <= 1e-4 else 'above'} the 1e-4 bar  ({dt:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
