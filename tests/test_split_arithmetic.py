"""The arithmetic of the opt-in f16 x 3 vocoder mode (indextts_amd/csrc/bigvgan_h3.hip), emulated on CPU over the reference-minted
BigVGAN fixtures: every conv of the oracle's forward is replaced by the split-operand sum  xh*wh + 2^-11 (xh*wl + xl*wh)  with f16
parts: a high part that is zero when it would be an f16 subnormal, and a low part pre-scaled by 2^11 (which then carries such values
whole), so the result does not depend on how the matrix pipe treats f16 subnormals.  This is what justified the kernel before it
was written and pins the choice of f16 (22 bits per pair) over bf16 parts (16 bits)."""
import os

import numpy as np
import pytest
import torch

from oracle import bigvgan_oracle as O


TINY = 6.103515625e-05                                           # smallest normal f16


def _split(t, dt, scale, flush):
    """flush: model a matrix pipe that reads f16 subnormals as zero"""
    h = t.to(dt).to(torch.float32)
    if dt == torch.float16 and scale != 1.0:
        h = torch.where(t.abs() < TINY, torch.zeros_like(h), h)   # the kernel's rule: no subnormal high parts
    lo = ((t - h) * scale).to(dt).to(torch.float32)
    if flush and dt == torch.float16:
        h = torch.where(h.abs() < TINY, torch.zeros_like(h), h)
        lo = torch.where(lo.abs() < TINY, torch.zeros_like(lo), lo)
    return h, lo / scale


def _run(monkeypatch, tag, golden_dir, dt, scale, flush):
    z = np.load(os.path.join(golden_dir, f"bigvgan_gen_{tag}.npz"))
    h = dict(O.V2_HPARAMS, upsample_initial_channel=int(z["upsample_initial_channel"]))
    sd = O.synth_weights(h, seed=int(z["seed"]), post_gain=float(z["post_gain"]))
    c1 = torch.nn.functional.conv1d

    def conv1d(x, w, b=None, **kw):
        if x.shape[1] < 32 or x.shape[1] != w.shape[0]:            # the mode covers the resblock convs (C_in == C_out, >= 32)
            return c1(x, w, b, **kw)
        xh, xl = _split(x, dt, scale, flush)
        wh, wl = _split(w, dt, scale, flush)
        y = c1(xh, wl, None, **kw) + c1(xl, wh, None, **kw)
        y = c1(xh, wh, None, **kw) + y
        return y if b is None else y + b.view(1, -1, 1)

    monkeypatch.setattr(O.F, "conv1d", conv1d)
    with torch.no_grad():
        wav = O.bigvgan_forward(sd, torch.from_numpy(z["mel"]), h).numpy()
    return float(np.sqrt(np.mean(np.square((wav - z["wav"]).astype(np.float64)))))


@pytest.mark.parametrize("tag", ["small", "loud"])
def test_f16_pair_with_scaled_low_part_is_f32_grade(monkeypatch, golden_dir, tag):
    err = _run(monkeypatch, tag, golden_dir, torch.float16, 2048.0, True)
    assert err <= 3e-6, err                                      # f32 oracle itself: 1.5e-7 (small) / 7.4e-7 (loud)


def test_bf16_pair_is_several_times_worse(monkeypatch, golden_dir):
    e16 = _run(monkeypatch, "loud", golden_dir, torch.float16, 2048.0, True)
    eb = _run(monkeypatch, "loud", golden_dir, torch.bfloat16, 1.0, False)
    assert eb > 3 * e16 and eb < 1e-4, (e16, eb)                 # 5e-6 vs 1.2e-6 (resblock convs only; 1.6e-5 with every conv split)


def test_unscaled_f16_low_part_fails_when_subnormals_flush(monkeypatch, golden_dir):
    err = _run(monkeypatch, "small", golden_dir, torch.float16, 1.0, True)
    assert err > 5e-5, err
