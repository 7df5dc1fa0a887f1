"""Prompt-audio front end on the HIP engine (the DSP half): host mirrors of the four callables the reference
pipeline applies to the speaker / emotion prompt before any network runs (indextts/infer_v2_5.py:626-648, 687-691):

    Resample(orig_freq, new_freq)(waveform)                      torchaudio.transforms.Resample               :627-628, :642
    mel_spectrogram(y, n_fft, num_mels, sampling_rate, ...)      indextts/s2mel/modules/audio.py:43-83        :266, :640
    fbank(waveform, num_mel_bins=80, dither=0, ...)              torchaudio.compliance.kaldi.fbank            :644-647
    SeamlessM4TFeatureExtractor()(audio, sampling_rate=16000)    transformers (w2v-bert-2.0 preprocessor)     :174, :631, :687
    MelSpectrogramFeatures()(audio)                              indextts/utils/feature_extractors.py:24-51   indextts/infer.py:318,535 (v1 / v1.5)

with the reference's argument names and result shapes.  The host side builds the constant tables in float64 (filter banks, windows,
FFT twiddles, the windowed-sinc resampling kernel) exactly as the published definitions give them and uploads them once per
configuration and device; the waveform work runs in `audio_kernels.hip` through the C ABI (`itts_resample_forward`,
`itts_fbank_forward`, `itts_tok_colnorm_forward`).  There is no CPU path: without the HIP library the calls raise.
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

FLT_EPSILON = 1.1920928955078125e-07
_tables: Dict[tuple, tuple] = {}         # (kind, parameters, device) -> device tensors; the reference keeps the same kind of global cache
                                         # (`mel_basis` / `hann_window` dicts, audio.py:39-40)


# ---- constant tables (host, float64) ------------------------------------------------------------------------------------------
def twiddles(n_fft: int) -> np.ndarray:
    """[n_fft / 2][2] = (cos, -sin)(2 pi m / n_fft): the forward-transform roots the FFT kernel multiplies by"""
    ang = 2.0 * np.pi * np.arange(n_fft // 2, dtype=np.float64) / n_fft
    return np.stack([np.cos(ang), -np.sin(ang)], axis=1).astype(np.float32)


def mel_basis_slaney(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax: Optional[float] = None) -> np.ndarray:
    """The bank `librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=)` defines (Slaney's auditory-toolbox scale: linear below 1 kHz,
    logarithmic above; triangles in Hz, each normalised to unit area) -> float32 [n_mels][1 + n_fft / 2]"""
    top = 0.5 * sr if fmax is None else float(fmax)
    lin_step, knee_hz = 200.0 / 3.0, 1000.0
    knee_mel, log_step = knee_hz / lin_step, math.log(6.4) / 27.0

    def to_mel(hz):
        return knee_mel + math.log(hz / knee_hz) / log_step if hz >= knee_hz else hz / lin_step

    pts = np.linspace(to_mel(float(fmin)), to_mel(top), n_mels + 2)
    edges = np.where(pts >= knee_mel, knee_hz * np.exp(log_step * (pts - knee_mel)), lin_step * pts)
    bins = np.linspace(0.0, 0.5 * sr, 1 + n_fft // 2)
    bank = np.zeros((n_mels, bins.size))
    for m in range(n_mels):
        lo, mid, hi = edges[m], edges[m + 1], edges[m + 2]
        rise, fall = (bins - lo) / (mid - lo), (hi - bins) / (hi - mid)
        bank[m] = np.clip(np.minimum(rise, fall), 0.0, None) * (2.0 / (hi - lo))
    return bank.astype(np.float32)


def mel_banks_kaldi(n_mels: int = 80, n_fft: int = 512, sr: float = 16000.0, low_freq: float = 20.0, high_freq: float = 0.0) -> np.ndarray:
    """Kaldi's mel bank (mel = 1127 ln(1 + f / 700), triangles in the MEL domain, n_mels + 2 equally spaced edges between low_freq and
    high_freq (<= 0: relative to Nyquist), evaluated at the first n_fft / 2 bin centres; the Nyquist column is zero)
    -> float32 [n_mels][n_fft / 2 + 1]"""
    to_mel = lambda hz: 1127.0 * np.log1p(np.asarray(hz, dtype=np.float64) / 700.0)
    hi = high_freq if high_freq > 0 else 0.5 * sr + high_freq
    m_lo, m_hi = float(to_mel(low_freq)), float(to_mel(hi))
    step = (m_hi - m_lo) / (n_mels + 1)
    bin_mel = to_mel(np.arange(n_fft // 2, dtype=np.float64) * (sr / n_fft))
    bank = np.zeros((n_mels, n_fft // 2 + 1))
    for m in range(n_mels):
        left, centre, right = m_lo + m * step, m_lo + (m + 1) * step, m_lo + (m + 2) * step
        bank[m, : n_fft // 2] = np.clip(np.minimum((bin_mel - left) / (centre - left), (right - bin_mel) / (right - centre)), 0.0, None)
    return bank.astype(np.float32)


def mel_banks_htk(sr: int, n_fft: int, n_mels: int, f_min: float = 0.0, f_max: Optional[float] = None) -> np.ndarray:
    """HTK-scale triangular bank without area normalisation (mel = 2595 log10(1 + f / 700); n_mels + 2 edges equally spaced in mel between
    f_min and f_max, triangles in Hz over the n_fft / 2 + 1 bin frequencies): what torchaudio's MelScale(norm=None, mel_scale="htk") applies
    -> float32 [n_mels][n_fft / 2 + 1]"""
    top = float(sr // 2) if f_max is None else float(f_max)
    mel_of = lambda hz: 2595.0 * math.log10(1.0 + hz / 700.0)
    edges_mel = np.linspace(mel_of(float(f_min)), mel_of(top), n_mels + 2)
    edges = 700.0 * (np.power(10.0, edges_mel / 2595.0) - 1.0)
    bins = np.linspace(0, sr // 2, n_fft // 2 + 1)
    bank = np.zeros((n_mels, bins.size))
    for m in range(n_mels):
        lo, mid, hi = edges[m], edges[m + 1], edges[m + 2]
        bank[m] = np.clip(np.minimum((bins - lo) / (mid - lo), (hi - bins) / (hi - mid)), 0.0, None)
    return bank.astype(np.float32)


def window_povey(n: int) -> np.ndarray:
    """Kaldi's default window: a symmetric Hann window raised to 0.85"""
    return np.power(0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / (n - 1)), 0.85).astype(np.float32)


def window_hann_periodic(n: int) -> np.ndarray:
    """The window the reference passes to torch.stft: torch.hann_window(n) (periodic), in torch's own float32 rounding"""
    return torch.hann_window(n, periodic=True, dtype=torch.float32).numpy()


def sinc_kernel(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> Tuple[np.ndarray, int, int, int]:
    """Hann-windowed sinc interpolation kernel of torchaudio's default resampler: one row of taps per output phase.
    -> (float32 [new'][2 width + orig'], width, orig', new') with the rates divided by their gcd"""
    g = math.gcd(int(orig), int(new))
    o, n = int(orig) // g, int(new) // g
    cutoff = min(o, n) * rolloff
    width = int(math.ceil(lowpass_filter_width * o / cutoff))
    tap = np.arange(-width, width + o, dtype=np.float64)[None, :] / o
    phase = -np.arange(n, dtype=np.float64)[:, None] / n
    t = np.clip((phase + tap) * cutoff, -lowpass_filter_width, lowpass_filter_width)
    hann = np.cos(t * math.pi / lowpass_filter_width / 2.0) ** 2
    t = t * math.pi
    sinc = np.where(t == 0.0, 1.0, np.sin(t) / np.where(t == 0.0, 1.0, t))
    return (sinc * hann * (cutoff / o)).astype(np.float32), width, o, n


def _cached(key, device, build):
    k = (key, str(device))
    if k not in _tables:
        _tables[k] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in build())
    return _tables[k]


def _as_rows(x, device) -> torch.Tensor:
    """-> contiguous float32 (rows, L) on `device`"""
    t = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x)
    t = t.detach().to(device=device, dtype=torch.float32)
    return t.reshape(-1, t.shape[-1]).contiguous()


def _device_of(x, device):
    if device is not None:
        return torch.device(device)
    if isinstance(x, torch.Tensor) and x.is_cuda:
        return x.device
    return torch.device("cuda", torch.cuda.current_device())


# ---- engine calls -----------------------------------------------------------------------------------------------------------------
def _fbank(rows: torch.Tensor, cfg: "_lib.FbankConfig", window: torch.Tensor, tw: torch.Tensor, mel: torch.Tensor) -> torch.Tensor:
    """rows (B, L) -> (B, frames, n_mels) for layout 0, (B, n_mels, frames) for layout 1"""
    L = _lib.lib()
    B, n = rows.shape
    frames = L.itts_fbank_frames(cfg, n)
    if frames < 0:
        raise ValueError("fbank: bad configuration")
    shape = (B, frames, cfg.n_mels) if cfg.layout == 0 else (B, cfg.n_mels, frames)
    out = torch.empty(shape, dtype=torch.float32, device=rows.device)
    if frames == 0:
        return out
    with _lib.on_device(rows.device):
        _lib.check(L.itts_fbank_forward(_lib.ptr(rows), B, n, rows.stride(0), cfg, _lib.ptr(window), _lib.ptr(tw), _lib.ptr(mel), _lib.ptr(out),
                                        shape[2], shape[1] * shape[2], _lib.stream_ptr(rows.device)), "itts_fbank_forward")
    return out


def _colnorm(x: torch.Tensor, out: torch.Tensor, mode: int, ddof: int = 0, eps: float = 0.0) -> torch.Tensor:
    """x (n, C) -> out[:n] (rows of out may be wider / more numerous than x's)"""
    L = _lib.lib()
    with _lib.on_device(x.device):
        _lib.check(L.itts_tok_colnorm_forward(_lib.ptr(x), _lib.ptr(out), x.shape[0], x.shape[1], out.stride(0), mode, ddof, float(eps),
                                              _lib.stream_ptr(x.device)), "itts_tok_colnorm_forward")
    return out


# ---- the reference callables ------------------------------------------------------------------------------------------------------
class Resample:
    """torchaudio.transforms.Resample(orig_freq, new_freq) with its defaults (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99)."""

    def __init__(self, orig_freq: int = 16000, new_freq: int = 16000, lowpass_filter_width: int = 6, rolloff: float = 0.99, device=None):
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        self.lowpass_filter_width, self.rolloff = int(lowpass_filter_width), float(rolloff)
        self.device = None if device is None else torch.device(device)

    def __call__(self, waveform: torch.Tensor) -> torch.Tensor:
        if self.orig_freq == self.new_freq:
            return waveform
        dev = _device_of(waveform, self.device)
        g = math.gcd(self.orig_freq, self.new_freq)
        o, n = self.orig_freq // g, self.new_freq // g
        width = int(math.ceil(self.lowpass_filter_width * o / (min(o, n) * self.rolloff)))
        (kd,) = _cached(("sinc", self.orig_freq, self.new_freq, self.lowpass_filter_width, self.rolloff), dev,
                        lambda: (sinc_kernel(self.orig_freq, self.new_freq, self.lowpass_filter_width, self.rolloff)[0],))
        assert kd.shape == (n, 2 * width + o)
        lead = tuple(waveform.shape[:-1])
        x = _as_rows(waveform, dev)
        length = x.shape[1]
        target = int(math.ceil(n * length / o))
        y = torch.empty(x.shape[0], target, dtype=torch.float32, device=dev)
        L = _lib.lib()
        with _lib.on_device(dev):
            _lib.check(L.itts_resample_forward(_lib.ptr(x), _lib.ptr(kd), _lib.ptr(y), x.shape[0], length, x.stride(0), target, y.stride(0), o, n,
                                               width, _lib.stream_ptr(dev)), "itts_resample_forward")
        return y.reshape(lead + (target,))

    forward = __call__


def mel_spectrogram(y: torch.Tensor, n_fft: int, num_mels: int, sampling_rate: int, hop_size: int, win_size: int, fmin: float,
                    fmax: Optional[float], center: bool = False, device=None) -> torch.Tensor:
    """indextts/s2mel/modules/audio.py:43-83.  y (B, L) in [-1, 1] -> log-mel (B, num_mels, frames), frames = 1 + (L + 2 pad - n_fft) // hop
    with pad = (n_fft - hop_size) // 2 of reflect padding (the pipeline passes center=False)."""
    if center:
        raise NotImplementedError("mel_spectrogram (HIP engine): center=False only, as the pipeline calls it (infer_v2_5.py:264)")
    if win_size != n_fft:
        raise NotImplementedError("mel_spectrogram (HIP engine): win_size == n_fft only (the shipped spect_params: 1024 / 1024)")
    dev = _device_of(y, device)
    window, tw, basis = _cached(("mel", n_fft, num_mels, sampling_rate, fmin, fmax), dev,
                                lambda: (window_hann_periodic(win_size), twiddles(n_fft), mel_basis_slaney(sampling_rate, n_fft, num_mels, fmin, fmax)))
    rows = _as_rows(y, dev)
    pad = int((n_fft - hop_size) / 2)
    if rows.shape[1] <= pad:
        raise ValueError(f"mel_spectrogram: reflect padding of {pad} needs more than {pad} samples (got {rows.shape[1]})")   # torch's pad raises too
    cfg = _lib.FbankConfig(frame_length=n_fft, hop=hop_size, n_fft=n_fft, n_mels=num_mels, pad=pad, remove_dc=0, power=1, take_log=1, layout=1,
                           preemphasis=0.0, mag_eps=1e-9, floor=1e-5, scale=1.0)
    return _fbank(rows, cfg, window, tw, basis)


class MelSpectrogramFeatures:
    """indextts/utils/feature_extractors.py:24-51 -- the conditioning mel of IndexTTS-1 / 1.5 (`MelSpectrogramFeatures()(audio)`,
    indextts/infer.py:318,535): torchaudio MelSpectrogram(power=1, HTK bank, no normalisation) + log(clip(., 1e-7)).
    audio (B, L) -> (B, n_mels, frames); "center": reflect padding n_fft / 2, frames = 1 + L // hop."""

    def __init__(self, sample_rate=24000, n_fft=1024, hop_length=256, win_length=None, n_mels=100, mel_fmin=0, mel_fmax=None, normalize=False,
                 padding="center", device=None):
        if padding not in ("center", "same"):
            raise ValueError("Padding must be 'center' or 'same'.")
        if normalize or (win_length not in (None, n_fft)):
            raise NotImplementedError("MelSpectrogramFeatures (HIP engine): normalize=False and win_length == n_fft (the pipeline's defaults)")
        self.sample_rate, self.n_fft, self.hop_length, self.n_mels = int(sample_rate), int(n_fft), int(hop_length), int(n_mels)
        self.mel_fmin, self.mel_fmax, self.padding = mel_fmin, mel_fmax, padding
        self.device = None if device is None else torch.device(device)

    def __call__(self, audio: torch.Tensor, **kwargs) -> torch.Tensor:
        dev = _device_of(audio, self.device)
        window, tw, bank = _cached(("htk", self.n_fft, self.n_mels, self.sample_rate, self.mel_fmin, self.mel_fmax), dev,
                                   lambda: (window_hann_periodic(self.n_fft), twiddles(self.n_fft),
                                            mel_banks_htk(self.sample_rate, self.n_fft, self.n_mels, self.mel_fmin, self.mel_fmax)))
        rows = _as_rows(audio, dev)
        pad = self.n_fft // 2 if self.padding == "center" else (self.n_fft - self.hop_length) // 2
        if rows.shape[1] <= pad:
            raise ValueError(f"MelSpectrogramFeatures: reflect padding of {pad} needs more than {pad} samples (got {rows.shape[1]})")
        cfg = _lib.FbankConfig(frame_length=self.n_fft, hop=self.hop_length, n_fft=self.n_fft, n_mels=self.n_mels, pad=pad, remove_dc=0, power=1,
                               take_log=1, layout=1, preemphasis=0.0, mag_eps=0.0, floor=1e-7, scale=1.0)
        return _fbank(rows, cfg, window, tw, bank)

    forward = __call__


def _kaldi_cfg(sample_frequency, frame_length, frame_shift, num_mel_bins, preemphasis_coefficient, remove_dc_offset, scale):
    win = int(sample_frequency * frame_length * 0.001)
    hop = int(sample_frequency * frame_shift * 0.001)
    n_fft = 1 << (win - 1).bit_length()                               # round_to_power_of_two
    return _lib.FbankConfig(frame_length=win, hop=hop, n_fft=n_fft, n_mels=num_mel_bins, pad=0, remove_dc=int(remove_dc_offset), power=2,
                            take_log=1, layout=0, preemphasis=float(preemphasis_coefficient), mag_eps=0.0, floor=FLT_EPSILON, scale=float(scale))


def fbank(waveform: torch.Tensor, num_mel_bins: int = 23, dither: float = 0.0, sample_frequency: float = 16000.0, frame_length: float = 25.0,
          frame_shift: float = 10.0, low_freq: float = 20.0, high_freq: float = 0.0, preemphasis_coefficient: float = 0.97,
          remove_dc_offset: bool = True, device=None, _scale: float = 1.0) -> torch.Tensor:
    """torchaudio.compliance.kaldi.fbank with its defaults (povey window, snip_edges, power spectrum, log, no energy column, no VTLN);
    waveform (channels, L): channel 0 -> (frames, num_mel_bins).  The pipeline calls it with num_mel_bins=80, dither=0."""
    if dither != 0.0:
        raise NotImplementedError("fbank (HIP engine): dither=0 only (the pipeline's setting; dither is random noise)")
    dev = _device_of(waveform, device)
    cfg = _kaldi_cfg(sample_frequency, frame_length, frame_shift, num_mel_bins, preemphasis_coefficient, remove_dc_offset, _scale)
    window, tw, banks = _cached(("kaldi", cfg.frame_length, cfg.n_fft, num_mel_bins, float(sample_frequency), low_freq, high_freq), dev,
                                lambda: (window_povey(cfg.frame_length), twiddles(cfg.n_fft),
                                         mel_banks_kaldi(num_mel_bins, cfg.n_fft, float(sample_frequency), low_freq, high_freq)))
    rows = _as_rows(waveform, dev)[:1]
    return _fbank(rows, cfg, window, tw, banks)[0]


class SeamlessM4TFeatureExtractor:
    """transformers' SeamlessM4TFeatureExtractor (default configuration: 80 mel bins, stride 2, 16 kHz, right padding with 0) ->
    {"input_features": (B, ceil(F / 2), 160), "attention_mask": (B, ceil(F / 2))} as torch tensors on the engine's device."""
    model_input_names = ["input_features", "attention_mask"]

    def __init__(self, feature_size: int = 80, sampling_rate: int = 16000, num_mel_bins: int = 80, padding_value: float = 0.0, stride: int = 2,
                 device=None, **kwargs):
        if (feature_size, sampling_rate, num_mel_bins, stride) != (80, 16000, 80, 2) or padding_value != 0.0:
            raise NotImplementedError("SeamlessM4TFeatureExtractor (HIP engine): the w2v-bert-2.0 preprocessor configuration only")
        self.feature_size, self.sampling_rate, self.num_mel_bins, self.padding_value, self.stride = 80, 16000, 80, 0.0, 2
        self.device = None if device is None else torch.device(device)

    @classmethod
    def from_pretrained(cls, *args, device=None, **kwargs):
        """The preprocessor has no learned state; the checkpoint directory's preprocessor_config.json holds the defaults above."""
        return cls(device=device)

    def __call__(self, raw_speech, sampling_rate: Optional[int] = None, return_tensors: Optional[str] = "pt", do_normalize_per_mel_bins: bool = True,
                 **kwargs) -> Dict[str, torch.Tensor]:
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(f"SeamlessM4TFeatureExtractor was built for {self.sampling_rate} Hz audio, got {sampling_rate}")
        if return_tensors not in (None, "pt"):
            raise NotImplementedError("SeamlessM4TFeatureExtractor (HIP engine): return_tensors='pt'")
        batched = isinstance(raw_speech, (list, tuple)) and len(raw_speech) > 0 and not np.isscalar(raw_speech[0])
        items: Sequence = list(raw_speech) if batched else [raw_speech]
        dev = _device_of(items[0], self.device)
        feats: List[torch.Tensor] = []
        for it in items:
            f = fbank(_as_rows(it, dev)[:1], num_mel_bins=80, dither=0.0, sample_frequency=16000.0, device=dev, _scale=32768.0)   # 2-D input: channel 0
            if f.shape[0] < 2:
                raise ValueError("SeamlessM4TFeatureExtractor: the audio is shorter than two 25 ms frames")
            feats.append(f)
        longest = max(f.shape[0] for f in feats)
        n_pad = longest + (-longest) % self.stride                    # padding=True, pad_to_multiple_of=2
        out = torch.zeros(len(feats), n_pad, 80, dtype=torch.float32, device=dev)
        mask = torch.zeros(len(feats), n_pad, dtype=torch.int32, device=dev)
        for b, f in enumerate(feats):
            if do_normalize_per_mel_bins:
                _colnorm(f, out[b], mode=1, ddof=1, eps=1e-7)
            else:
                out[b, : f.shape[0]] = f
            mask[b, : f.shape[0]] = 1
        return {"input_features": out.reshape(len(feats), n_pad // self.stride, 80 * self.stride),
                "attention_mask": mask[:, 1::self.stride].contiguous()}


def subtract_mean(feat: torch.Tensor) -> torch.Tensor:
    """`feat - feat.mean(dim=0, keepdim=True)` (infer_v2_5.py:648) on the engine; feat (frames, C)"""
    x = feat.contiguous()
    return _colnorm(x, torch.empty_like(x), mode=0)
