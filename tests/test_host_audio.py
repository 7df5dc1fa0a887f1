"""CPU: host side of the prompt-audio front end (indextts_amd/audio.py) -- the constant tables it uploads (filter banks, windows, FFT
twiddles, resampling kernel) against the oracle's independent construction and transformers' implementations, and a float32 numpy
EMULATION of audio_kernels.hip's frame pipeline (same framing rule, same radix-2 Stockham index arithmetic, same table layout) against
the golden features: this pins the algorithm the kernel implements and shows the float32 error budget the GPU test then holds the
kernel to.  The kernel itself only runs on the GPU (tests/test_gpu_audio.py)."""
import os

import numpy as np
import pytest
import torch

from indextts_amd import _lib, audio
from oracle import audio_oracle as AO
from tools.make_golden_audio import LENGTHS_16K, LENGTHS_22K


def emulate_fbank(wave, cfg, window, tw, mel):
    """float32 numpy model of fbank_kernel: wave (L,) -> (frames, n_mels)"""
    f32 = np.float32
    N, FL = cfg.n_fft, cfg.frame_length
    n = wave.size
    frames = 1 + (n + 2 * cfg.pad - FL) // cfg.hop
    idx = np.arange(frames)[:, None] * cfg.hop - cfg.pad + np.arange(FL)[None, :]
    idx = np.where(idx < 0, -idx, idx)
    idx = np.where(idx >= n, 2 * (n - 1) - idx, idx)
    raw = wave.astype(f32)[idx] * f32(cfg.scale)
    mean = raw.sum(1, dtype=f32, keepdims=True) / f32(FL) if cfg.remove_dc else np.zeros((frames, 1), f32)
    cur = raw - mean
    prev = np.concatenate([cur[:, :1], cur[:, :-1]], 1)
    buf = np.zeros((frames, N), dtype=np.complex64)
    buf[:, :FL] = ((cur - f32(cfg.preemphasis) * prev) * window.astype(f32)).astype(f32)
    twc = (tw[:, 0] + 1j * tw[:, 1]).astype(np.complex64)
    half = N // 2
    j = np.arange(half)
    Ns = 1
    while Ns < N:
        k = j & (Ns - 1)
        w = twc[k * (half // Ns)]
        u0, u1 = buf[:, j], (buf[:, j + half] * w).astype(np.complex64)
        j0 = ((j - k) << 1) + k
        nxt = np.empty_like(buf)
        nxt[:, j0], nxt[:, j0 + Ns] = u0 + u1, u0 - u1
        buf = nxt
        Ns <<= 1
    spec = buf[:, : half + 1]
    p = (spec.real * spec.real + spec.imag * spec.imag).astype(f32)
    P = p if cfg.power == 2 else np.sqrt(p + f32(cfg.mag_eps), dtype=f32)
    m = np.maximum((P @ mel.T.astype(f32)).astype(f32), f32(cfg.floor))
    return np.log(m) if cfg.take_log else m


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "audio.npz"))


def test_tables_match_oracle_and_transformers():
    from transformers.audio_utils import mel_filter_bank, window_function
    assert np.abs(audio.mel_basis_slaney(22050, 1024, 80, 0, None) - AO.slaney_mel_basis(22050, 1024, 80, 0, None)).max() <= 1e-9
    assert np.abs(audio.mel_basis_slaney(16000, 512, 40, 50.0, 7000.0) - AO.slaney_mel_basis(16000, 512, 40, 50.0, 7000.0)).max() <= 1e-9
    kb = mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20, max_frequency=8000, sampling_rate=16000, norm=None,
                         mel_scale="kaldi", triangularize_in_mel_space=True).T
    assert np.abs(audio.mel_banks_kaldi() - kb).max() <= 1e-7
    assert np.abs(audio.window_povey(400) - window_function(400, "povey", periodic=False)).max() <= 1e-7
    assert np.array_equal(audio.window_hann_periodic(1024), torch.hann_window(1024).numpy())
    hb = mel_filter_bank(num_frequency_bins=513, num_mel_filters=100, min_frequency=0.0, max_frequency=12000.0, sampling_rate=24000, norm=None,
                         mel_scale="htk").T
    assert np.abs(audio.mel_banks_htk(24000, 1024, 100) - hb).max() <= 1e-7 and np.abs(AO.htk_mel_banks(24000, 1024, 100) - hb).max() <= 1e-7
    tw = audio.twiddles(512)
    ref = np.exp(-2j * np.pi * np.arange(256) / 512)
    assert np.abs(tw[:, 0] + 1j * tw[:, 1] - ref).max() <= 1e-7
    for orig, new in ((24000, 22050), (44100, 16000), (22050, 16000), (16000, 24000)):
        k, width, o, n = audio.sinc_kernel(orig, new)
        ko, wo, oo, no = AO.sinc_resample_kernel(orig, new)
        assert (width, o, n) == (wo, oo, no) and k.shape == tuple(ko.shape) == (n, 2 * width + o)
        assert np.abs(k - ko.numpy()).max() <= 1e-7


def test_fbank_config_mirrors_header():
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "include", "indextts_hip.h")).read()
    body = src[src.index("typedef struct {\n    int32_t frame_length;"): src.index("} itts_fbank_config;")]
    fields = re.findall(r"^\s+(?:int32_t|float)\s+(\w+);", body, flags=re.M)
    assert fields == [f for f, _ in _lib.FbankConfig._fields_]
    # the struct crosses the real C ABI (host-only entry point: no GPU needed)
    L = _lib.lib()
    kal = audio._kaldi_cfg(16000.0, 25.0, 10.0, 80, 0.97, True, 1.0)
    assert [L.itts_fbank_frames(kal, n) for n in (399, 400, 559, 560, 20817, 33040)] == [0, 1, 1, 2, 128, 205]
    mel = _lib.FbankConfig(frame_length=1024, hop=256, n_fft=1024, n_mels=80, pad=384, remove_dc=0, power=1, take_log=1, layout=1)
    assert [L.itts_fbank_frames(mel, n) for n in (19000, 30001)] == [74, 117]
    assert L.itts_fbank_frames(_lib.FbankConfig(frame_length=0, hop=1), 100) < 0


def test_kernel_algorithm_emulation_vs_goldens(gold):
    # SeamlessM4T / Kaldi: 400-sample frames in a 512-point transform, snip-edges framing
    cfg = audio._kaldi_cfg(16000.0, 25.0, 10.0, 80, 0.97, True, 32768.0)
    assert (cfg.frame_length, cfg.hop, cfg.n_fft) == (400, 160, 512)
    tabs = (audio.window_povey(400), audio.twiddles(512), audio.mel_banks_kaldi())
    for i, n in enumerate(LENGTHS_16K):
        x = gold[f"wave16k_{i}"]
        f = emulate_fbank(x, cfg, *tabs)
        assert f.shape == (1 + (n - 400) // 160, 80)
        z = (f - f.mean(0, keepdims=True)) / np.sqrt(f.var(0, ddof=1, keepdims=True) + 1e-7)
        ref = gold[f"seamless_feat_{i}"].reshape(-1, 80)[: f.shape[0]]
        err = np.abs(z - ref).max()
        print(f"float32 kernel model vs transformers' float64 features, n={n}: max|d| {err:.2e}")
        assert err <= 5e-3
        cfg1 = audio._kaldi_cfg(16000.0, 25.0, 10.0, 80, 0.97, True, 1.0)
        k = emulate_fbank(x, cfg1, *tabs)
        anchor = gold[f"kaldi_anchor_{i}"]
        loud = anchor > anchor.max() - 12.0
        assert np.abs(k - anchor)[loud].max() <= 2e-4 and np.abs(k - anchor).max() <= 1e-2
    # log-mel of the 22 kHz prompt: 1024-point frames, reflect padding 384
    cfg = _lib.FbankConfig(frame_length=1024, hop=256, n_fft=1024, n_mels=80, pad=384, remove_dc=0, power=1, take_log=1, layout=1, preemphasis=0.0,
                           mag_eps=1e-9, floor=1e-5, scale=1.0)
    tabs = (audio.window_hann_periodic(1024), audio.twiddles(1024), audio.mel_basis_slaney(22050, 1024, 80, 0, None))
    for i, n in enumerate(LENGTHS_22K):
        m = emulate_fbank(gold[f"wave22k_{i}"], cfg, *tabs).T
        err = np.abs(m - gold[f"refmel_{i}"]).max()
        print(f"float32 kernel model vs the reference mel_spectrogram, n={n}: max|d| {err:.2e}")
        assert m.shape == gold[f"refmel_{i}"].shape and err <= 2e-4


def test_calls_fail_loudly_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        audio.fbank(torch.zeros(1, 16000), num_mel_bins=80)
    with pytest.raises(NotImplementedError):
        audio.mel_spectrogram(torch.zeros(1, 4000), 1024, 80, 22050, 256, 1024, 0, None, center=True)
    with pytest.raises(NotImplementedError):
        audio.fbank(torch.zeros(1, 16000), dither=1.0)
    assert audio.Resample(16000, 16000)(torch.ones(3)).shape == (3,)


class _EmulatedLibrary:
    """Stands in for libindextts_hip.so's three audio entry points on CPU tensors (the float32 numpy model above), so the HOST code of
    indextts_amd/audio.py -- argument marshalling, strides, batching, padding, masks -- runs here.  Test infrastructure only."""

    @staticmethod
    def _arr(p, shape):
        import ctypes as C
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n,)).reshape(shape)

    def itts_fbank_frames(self, cfg, n):
        padded = n + 2 * cfg.pad
        return 0 if padded < cfg.frame_length else 1 + (padded - cfg.frame_length) // cfg.hop

    def itts_fbank_forward(self, wave, B, n, wstride, cfg, window, tw, mel, out, ld_out, ostride, stream):
        frames = self.itts_fbank_frames(cfg, n)
        w = self._arr(wave, ((B - 1) * wstride + n,))
        o = self._arr(out, ((B - 1) * ostride + (frames if cfg.layout == 0 else cfg.n_mels) * ld_out,))
        tabs = (self._arr(window, (cfg.frame_length,)), self._arr(tw, (cfg.n_fft // 2, 2)), self._arr(mel, (cfg.n_mels, cfg.n_fft // 2 + 1)))
        for b in range(B):
            f = emulate_fbank(w[b * wstride: b * wstride + n], cfg, *tabs)
            blk = o[b * ostride:]
            if cfg.layout == 0:
                blk[: frames * ld_out].reshape(frames, ld_out)[:, : cfg.n_mels] = f
            else:
                blk[: cfg.n_mels * ld_out].reshape(cfg.n_mels, ld_out)[:, :frames] = f.T
        return 0

    def itts_resample_forward(self, x, kern, y, B, L_in, xs, L_out, ys, orig, new, width, stream):
        taps = 2 * width + orig
        k = self._arr(kern, (new, taps))
        xa, ya = self._arr(x, ((B - 1) * xs + L_in,)), self._arr(y, ((B - 1) * ys + L_out,))
        idx = np.arange(L_out)
        src = (idx // new)[:, None] * orig - width + np.arange(taps)[None, :]
        ok = (src >= 0) & (src < L_in)
        for b in range(B):
            row = xa[b * xs: b * xs + L_in]
            ya[b * ys: b * ys + L_out] = (np.where(ok, row[np.clip(src, 0, L_in - 1)], 0.0) * k[idx % new]).sum(1, dtype=np.float32)
        return 0

    def itts_tok_colnorm_forward(self, x, out, n, C, ld_out, mode, ddof, eps, stream):
        xa = self._arr(x, (n, C))
        o = self._arr(out, (n, ld_out))
        d = xa - xa.mean(0, keepdims=True)
        o[:, :C] = d / np.sqrt(xa.var(0, ddof=ddof, keepdims=True) + eps) if mode == 1 else d
        return 0


def test_host_paths_on_an_emulated_library(gold, monkeypatch):
    """audio.py's host code end to end on CPU tensors with the library's audio entry points replaced by the numpy model."""
    import math
    monkeypatch.setattr(_lib, "lib", lambda: _EmulatedLibrary())
    monkeypatch.setattr(_lib, "stream_ptr", lambda device=None: None)
    monkeypatch.setattr(audio, "_device_of", lambda x, device: torch.device("cpu"))
    monkeypatch.setattr(audio, "_tables", {})
    fe = audio.SeamlessM4TFeatureExtractor.from_pretrained("unused")
    o = fe([gold["wave16k_0"], torch.from_numpy(gold["wave16k_1"])[None]], sampling_rate=16000)
    f, m = o["input_features"].numpy(), o["attention_mask"].numpy()
    n0, n1 = gold["seamless_feat_0"].shape[1], gold["seamless_feat_1"].shape[1]
    assert f.shape == (2, n1, 160) and m.dtype == np.int32 and m[0].sum() == n0 and np.array_equal(m[1:], gold["seamless_mask_1"])
    assert np.abs(f[0, :n0] - gold["seamless_feat_0"][0]).max() <= 5e-3 and np.abs(f[0, n0:]).max() == 0.0
    assert np.abs(f[1] - gold["seamless_feat_1"][0]).max() <= 5e-3
    kw = dict(n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024, fmin=0, fmax=None, center=False)
    a, b = gold["wave22k_0"], gold["wave22k_1"][: gold["wave22k_0"].size]
    both = audio.mel_spectrogram(torch.from_numpy(np.stack([a, b])), **kw)
    assert both.shape == (2, 80, gold["refmel_0"].shape[1]) and float((both[0] - torch.from_numpy(gold["refmel_0"])).abs().max()) <= 2e-4
    assert float((both[1] - AO.mel_spectrogram(torch.from_numpy(b)[None])[0]).abs().max()) <= 2e-4
    k = audio.fbank(torch.from_numpy(gold["wave16k_0"])[None], num_mel_bins=80, dither=0, sample_frequency=16000)
    assert k.shape == gold["kaldi_anchor_0"].shape and float((audio.subtract_mean(k) - (k - k.mean(0, keepdim=True))).abs().max()) <= 1e-5
    assert audio.fbank(torch.zeros(1, 399), num_mel_bins=80).shape == (0, 80)
    x = torch.from_numpy(gold["wave22k_1"])
    for orig, new in ((24000, 22050), (44100, 16000)):
        y = audio.Resample(orig, new)(x[None])
        ref = AO.resample(x[None], orig, new)
        assert y.shape == ref.shape == (1, math.ceil(new * x.numel() / orig)) and float((y - ref).abs().max()) <= 1e-5
    two = torch.stack([x[:5000], x[5000:10000]])
    assert float((audio.Resample(24000, 16000)(two) - AO.resample(two, 24000, 16000)).abs().max()) <= 1e-5
    # v1 / v1.5 conditioning mel: centred reflect padding, 100 HTK bins, magnitude, log clip 1e-7
    for padding in ("center", "same"):
        m = audio.MelSpectrogramFeatures(padding=padding)(two)
        ref = AO.mel_spectrogram_features(two, padding=padding)
        assert m.shape == ref.shape == (2, 100, 1 + 5000 // 256 if padding == "center" else 1 + (5000 + 768 - 1024) // 256)
        assert float((m - ref).abs().max()) <= 5e-4
