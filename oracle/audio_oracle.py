"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the prompt-audio front end that feeds the speaker bundle (SURVEY.md section 8 f-3,
the DSP half: indextts/infer_v2_5.py:626-648).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.

What it restates, and what pins each piece:
  resample(orig -> new)          torchaudio.transforms.Resample (infer_v2_5.py:627-628,642), defaults sinc_interp_hann, lowpass_filter_width 6,
                                 rolloff 0.99.  torchaudio==2.8.* is a third-party dependency that is NOT installed here and not under
                                 /root/reference: the published algorithm (torchaudio/functional/functional.py `_get_sinc_resample_kernel`,
                                 `_apply_sinc_resample_kernel`) is restated.  **Parity unpinned**; tests hold it to DSP properties instead
                                 (a band-limited sine resamples to the analytic sine, identity at equal rates, output length rule).
  mel_spectrogram                indextts/s2mel/modules/audio.py:43-83 (reflect pad, torch.stft with a periodic Hann window, magnitude with
                                 1e-9 inside the root, Slaney mel basis, log of clamp 1e-5).  Pinned to the reference function itself, run
                                 here with `librosa.filters.mel` stubbed by `slaney_mel_basis` below (librosa==0.10.2.post1 is absent); the
                                 mel basis is pinned separately to transformers' `mel_filter_bank(norm="slaney", mel_scale="slaney")`, the
                                 librosa-compatible bank behind its Whisper feature extractor.
  kaldi_fbank                    torchaudio.compliance.kaldi.fbank(num_mel_bins=80, dither=0, sample_frequency=16000) (infer_v2_5.py:644-647):
                                 snip_edges framing, DC removal, pre-emphasis 0.97 with a replicated first sample, Povey window, 512-point
                                 power spectrum, Kaldi mel banks (20 Hz .. Nyquist), log(max(., eps_f32)).  torchaudio is absent: restated from
                                 the published source and anchored on transformers' Kaldi-compatible `spectrogram` + `mel_filter_bank(
                                 mel_scale="kaldi", triangularize_in_mel_space=True)` with the same parameters (tests/test_oracle_audio.py).
  seamless_features              transformers' SeamlessM4TFeatureExtractor (infer_v2_5.py:174,631): the same Kaldi fbank on the waveform
                                 scaled by 2^15, per-mel-bin mean / unbiased-variance normalisation, zero padding to an even frame count,
                                 pairs of frames stacked to 160 features, attention mask of the odd frames.  Pinned to the installed
                                 transformers class (exact algorithm, float64 inside like its numpy code).
  mel_spectrogram_features       indextts/utils/feature_extractors.py:24-51 (`MelSpectrogramFeatures`, the conditioning mel of IndexTTS-1 / 1.5,
                                 indextts/infer.py:318,535): torchaudio.transforms.MelSpectrogram(24 kHz, n_fft 1024, hop 256, power 1, 100 HTK
                                 mel bins without normalisation, centred reflect padding) + safe_log (clip 1e-7).  torchaudio is absent: the
                                 transform is restated from its published definition (Spectrogram = torch.stft with a periodic Hann window;
                                 MelScale = `melscale_fbanks(norm=None, mel_scale="htk")`), the bank anchored on transformers'
                                 `mel_filter_bank(norm=None, mel_scale="htk")`.  **Parity unpinned** beyond that anchor.
Golden vectors: tests/golden/audio.npz (tools/make_golden_audio.py).
"""
import math

import numpy as np
import torch

EPS_F32 = 1.1920928955078125e-07


# ---- filter banks and windows ---------------------------------------------------------------------------------------------
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=) with its defaults htk=False, norm='slaney' -> float32 [n_mels][1 + n_fft/2]"""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, fftfreqs.size))
    for i in range(n_mels):
        w[i] = np.maximum(0.0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def htk_mel_banks(sr, n_fft, n_mels, f_min=0.0, f_max=None):
    """torchaudio.functional.melscale_fbanks(n_fft // 2 + 1, f_min, f_max, n_mels, sr, norm=None, mel_scale="htk"), transposed
    -> float32 [n_mels][n_fft / 2 + 1]"""
    f_max = float(sr // 2) if f_max is None else float(f_max)
    to_mel = lambda f: 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
    to_hz = lambda m: 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)
    all_freqs = np.linspace(0, sr // 2, n_fft // 2 + 1)
    f_pts = to_hz(np.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2))
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down, up = -slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).T.astype(np.float32)


def kaldi_mel_banks(n_mels=80, n_fft=512, sr=16000.0, low=20.0, high=0.0):
    """torchaudio.compliance.kaldi.get_mel_banks (no VTLN) + the zero Nyquist column fbank() appends -> float64 [n_mels][n_fft/2 + 1]"""
    mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
    nyq = 0.5 * sr
    if high <= 0:
        high += nyq
    nb = n_fft // 2
    m_lo, m_hi = mel(low), mel(high)
    delta = (m_hi - m_lo) / (n_mels + 1)
    b = np.arange(n_mels, dtype=np.float64)[:, None]
    left, center, right = m_lo + b * delta, m_lo + (b + 1) * delta, m_lo + (b + 2) * delta
    m = mel(sr / n_fft * np.arange(nb, dtype=np.float64))[None, :]
    banks = np.maximum(0.0, np.minimum((m - left) / (center - left), (right - m) / (right - center)))
    return np.concatenate([banks, np.zeros((n_mels, 1))], axis=1)


def povey_window(n=400):
    """hann(periodic=False) ** 0.85 -> float64 [n]"""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / (n - 1))) ** 0.85


# ---- resampling -------------------------------------------------------------------------------------------------------------
def sinc_resample_kernel(orig, new, lowpass_filter_width=6, rolloff=0.99):
    """-> (float32 kernel [new/g][2 width + orig/g], width, orig/g, new/g)"""
    g = math.gcd(int(orig), int(new))
    o, n = int(orig) // g, int(new) // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = torch.arange(-width, width + o, dtype=torch.float64)[None, :] / o
    t = torch.arange(0, -n, -1, dtype=torch.float64)[:, None] / n + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    k = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / o)
    return k.to(torch.float32), width, o, n


def resample(wave: torch.Tensor, orig: int, new: int) -> torch.Tensor:
    """wave (..., L) float32 -> (..., ceil(new L / orig))"""
    if int(orig) == int(new):
        return wave
    k, width, o, n = sinc_resample_kernel(orig, new)
    shape = wave.shape
    x = wave.reshape(-1, shape[-1]).float()
    L = x.shape[-1]
    x = torch.nn.functional.pad(x, (width, width + o))
    y = torch.nn.functional.conv1d(x[:, None], k[:, None], stride=o)            # (rows, n, blocks)
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = int(math.ceil(n * L / o))
    return y[..., :target].reshape(shape[:-1] + (target,))


# ---- features -----------------------------------------------------------------------------------------------------------------
def mel_spectrogram(y: torch.Tensor, n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024, fmin=0, fmax=None,
                    center=False) -> torch.Tensor:
    """y (B, L) float32 -> (B, num_mels, frames) log-mel, indextts/s2mel/modules/audio.py:43-83"""
    basis = torch.from_numpy(slaney_mel_basis(sampling_rate, n_fft, num_mels, fmin, fmax))
    pad = int((n_fft - hop_size) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.view_as_real(torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=torch.hann_window(win_size),
                                         center=center, pad_mode="reflect", normalized=False, onesided=True, return_complex=True))
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(basis, spec), min=1e-5))


def mel_spectrogram_features(audio: torch.Tensor, sample_rate=24000, n_fft=1024, hop_length=256, n_mels=100, mel_fmin=0, mel_fmax=None,
                             padding="center") -> torch.Tensor:
    """audio (B, L) -> (B, n_mels, frames) log-mel, indextts/utils/feature_extractors.py:24-51"""
    if padding == "same":
        pad = n_fft - hop_length
        audio = torch.nn.functional.pad(audio.unsqueeze(1), (pad // 2, pad // 2), mode="reflect").squeeze(1)
    spec = torch.stft(audio, n_fft, hop_length=hop_length, win_length=n_fft, window=torch.hann_window(n_fft), center=padding == "center",
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs()
    mel = torch.matmul(torch.from_numpy(htk_mel_banks(sample_rate, n_fft, n_mels, mel_fmin, mel_fmax)), spec)
    return torch.log(torch.clip(mel, min=1e-7))


def kaldi_fbank(wave: torch.Tensor, num_mel_bins=80, sample_frequency=16000.0, frame_length_ms=25.0, frame_shift_ms=10.0,
                preemphasis=0.97, dtype=torch.float32) -> torch.Tensor:
    """wave (1, L) or (L,) -> (frames, num_mel_bins).  torchaudio computes in the input dtype (float32 in the pipeline); float64 is
    offered for the precision audit of the engine kernel."""
    x = wave.reshape(-1, wave.shape[-1])[0].to(dtype)
    win = int(sample_frequency * frame_length_ms * 0.001)
    shift = int(sample_frequency * frame_shift_ms * 0.001)
    n_fft = 1 << (win - 1).bit_length()
    if x.numel() < win:
        return torch.empty(0, num_mel_bins, dtype=dtype)
    m = 1 + (x.numel() - win) // shift
    fr = x.as_strided((m, win), (shift, 1)).clone()
    fr = fr - fr.mean(dim=1, keepdim=True)
    prev = torch.cat([fr[:, :1], fr[:, :-1]], dim=1)                           # replicate pad on the left
    fr = fr - preemphasis * prev
    fr = fr * torch.from_numpy(povey_window(win)).to(dtype)
    fr = torch.nn.functional.pad(fr, (0, n_fft - win))
    power = torch.fft.rfft(fr).abs().pow(2.0)
    banks = torch.from_numpy(kaldi_mel_banks(num_mel_bins, n_fft, sample_frequency)).to(dtype)
    mel = power @ banks.T
    return torch.max(mel, torch.tensor(EPS_F32, dtype=dtype)).log()


def seamless_features(wave: np.ndarray, stride=2):
    """wave (L,) float32 at 16 kHz -> (input_features float32 (1, ceil(F / 2), 160), attention_mask int32 (1, ceil(F / 2))).

    transformers SeamlessM4TFeatureExtractor.__call__ for one utterance (padding=True, pad_to_multiple_of=2): float64 framing like
    `audio_utils.spectrogram`, the spectrum rounded to complex64 before the power (as its `np.empty(..., complex64)` buffer does)."""
    w = np.asarray(wave, dtype=np.float32).astype(np.float64) * (2 ** 15)
    win, hop, n_fft = 400, 160, 512
    F = int(1 + np.floor((w.size - win) / hop))
    window = povey_window(win)
    banks = kaldi_mel_banks(80, n_fft, 16000.0).astype(np.float64)
    spec = np.empty((F, n_fft // 2 + 1), dtype=np.complex64)
    buf = np.zeros(n_fft)
    for i in range(F):
        fr = w[i * hop: i * hop + win].copy()
        fr -= fr.mean()
        fr[1:] -= 0.97 * fr[:-1].copy()
        fr[0] *= 1 - 0.97
        buf[:win] = fr * window
        spec[i] = np.fft.rfft(buf)
    power = np.abs(spec, dtype=np.float64) ** 2.0
    feats = np.log(np.maximum(EPS_F32, banks @ power.T)).T.astype(np.float32)           # (F, 80)
    feats = (feats - feats.mean(0, keepdims=True)) / np.sqrt(feats.var(0, ddof=1, keepdims=True) + 1e-7)
    mask = np.ones(F, dtype=np.int32)
    if F % stride:
        feats = np.concatenate([feats, np.zeros((stride - F % stride, 80), dtype=feats.dtype)], 0)
        mask = np.concatenate([mask, np.zeros(stride - F % stride, dtype=np.int32)])
    n = feats.shape[0]
    return feats.reshape(1, n // stride, 80 * stride).astype(np.float32), mask[np.arange(n) % stride == 1][None]
