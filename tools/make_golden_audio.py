"""Mint tests/golden/audio.npz: outputs of the code the reference pipeline itself runs for the prompt-audio front end
(indextts/infer_v2_5.py:626-648), executed HERE on seeded waveforms.

  seamless_*   transformers' own SeamlessM4TFeatureExtractor (the class infer_v2_5.py:174 loads), default configuration.
  refmel_*     the reference's own `mel_spectrogram` (indextts/s2mel/modules/audio.py, loaded from /root/reference) with the pipeline's
               arguments (n_fft 1024, win 1024, hop 256, 80 mels, 22050 Hz, fmin 0, fmax None).  Its `librosa.filters.mel` import is
               served by oracle/audio_oracle.slaney_mel_basis (librosa is not installed); that basis is checked here against
               transformers' librosa-compatible `mel_filter_bank(norm="slaney", mel_scale="slaney")`.
  kaldi_*      torchaudio is not installed, so there is no reference output for `kaldi.fbank`; the stored anchor is transformers'
               Kaldi-compatible `spectrogram` (float64 inside) with fbank's parameters on the UNSCALED waveform, which the float32
               restatement must track to float32 rounding.
The oracle and the engine are tested against these arrays (tests/test_oracle_audio.py, tests/test_gpu_audio.py).
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import audio_oracle as AO  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
LENGTHS_16K = (20817, 33040)         # 128 (even) and 205 (odd) Kaldi frames
LENGTHS_22K = (19000, 30001)


def speechlike(n, sr, seed):
    """Seeded test signal with a wide dynamic range across frequency and time: harmonic stack with vibrato under a slow envelope,
    a chirp, and low-level noise; peak ~0.6."""
    g = np.random.RandomState(seed)
    t = np.arange(n) / sr
    f0 = 110 + 40 * np.sin(2 * np.pi * 1.3 * t)
    ph = 2 * np.pi * np.cumsum(f0) / sr
    x = sum(np.sin(k * ph) / k ** 1.2 for k in range(1, 24))
    x *= 0.15 * (0.55 + 0.45 * np.sin(2 * np.pi * 2.1 * t + 0.4)) ** 2
    x += 0.05 * np.sin(2 * np.pi * (300 * t + 0.5 * (0.42 * sr / t[-1]) * t * t))
    x += 0.003 * g.randn(n)
    x[: n // 17] *= 0.01                                                    # a near-silent lead-in
    return (x / np.abs(x).max() * 0.6).astype(np.float32)


def load_reference_mel():
    lib = types.ModuleType("librosa")
    filt = types.ModuleType("librosa.filters")
    filt.mel = lambda sr, n_fft, n_mels, fmin, fmax: AO.slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax)
    lib.filters = filt
    sys.modules["librosa"], sys.modules["librosa.filters"] = lib, filt
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_audio", "/root/reference/indextts/s2mel/modules/audio.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.mel_spectrogram


def main():
    from transformers import SeamlessM4TFeatureExtractor
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    out = {}
    # filter banks / window of the restatement against transformers' implementations of the same published definitions
    tb = mel_filter_bank(num_frequency_bins=513, num_mel_filters=80, min_frequency=0.0, max_frequency=11025.0, sampling_rate=22050,
                         norm="slaney", mel_scale="slaney").T
    print("slaney basis vs transformers' librosa-compatible bank: max|d| %.2e (max %.3f)" %
          (np.abs(AO.slaney_mel_basis(22050, 1024, 80, 0, None) - tb).max(), tb.max()))
    kb = mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20, max_frequency=8000, sampling_rate=16000, norm=None,
                         mel_scale="kaldi", triangularize_in_mel_space=True)
    print("kaldi banks vs transformers: max|d| %.2e" % np.abs(AO.kaldi_mel_banks().T - kb).max())
    pw = window_function(400, "povey", periodic=False)
    print("povey window vs transformers: max|d| %.2e" % np.abs(AO.povey_window() - pw).max())

    fe = SeamlessM4TFeatureExtractor()
    for i, n in enumerate(LENGTHS_16K):
        x = speechlike(n, 16000, 40 + i)
        o = fe(x, sampling_rate=16000, return_tensors="np")
        mine, mask = AO.seamless_features(x)
        print(f"seamless n={n}: features {o['input_features'].shape}, mask sum {int(o['attention_mask'].sum())}, oracle vs transformers "
              f"max|d| {np.abs(mine - o['input_features']).max():.2e}, mask equal {np.array_equal(mask, o['attention_mask'])}")
        out[f"wave16k_{i}"], out[f"seamless_feat_{i}"], out[f"seamless_mask_{i}"] = x, o["input_features"], o["attention_mask"]
        anchor = spectrogram(x.astype(np.float64), pw, frame_length=400, hop_length=160, fft_length=512, power=2.0, center=False,
                             preemphasis=0.97, mel_filters=kb, log_mel="log", mel_floor=AO.EPS_F32, remove_dc_offset=True).T
        k32 = AO.kaldi_fbank(torch.from_numpy(x)[None]).numpy()
        k64 = AO.kaldi_fbank(torch.from_numpy(x)[None], dtype=torch.float64).numpy()
        print(f"kaldi fbank n={n}: {k32.shape}, float64 restatement vs transformers' Kaldi-compatible spectrogram max|d| "
              f"{np.abs(k64 - anchor).max():.2e}; float32 restatement vs it {np.abs(k32 - anchor).max():.2e} (range {anchor.min():.1f}..{anchor.max():.1f})")
        out[f"kaldi_anchor_{i}"] = anchor.astype(np.float32)
    ref_mel = load_reference_mel()
    for i, n in enumerate(LENGTHS_22K):
        x = speechlike(n, 22050, 50 + i)
        with torch.no_grad():
            r = ref_mel(torch.from_numpy(x)[None], n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024, fmin=0,
                        fmax=None, center=False)
            mine = AO.mel_spectrogram(torch.from_numpy(x)[None])
        print(f"mel_spectrogram n={n}: {tuple(r.shape)}, oracle vs the reference function max|d| {float((mine - r).abs().max()):.2e} "
              f"(range {float(r.min()):.2f}..{float(r.max()):.2f})")
        out[f"wave22k_{i}"], out[f"refmel_{i}"] = x, r[0].numpy()
    np.savez_compressed(os.path.join(GOLD, "audio.npz"), **out)
    print("wrote audio.npz", os.path.getsize(os.path.join(GOLD, "audio.npz")), "bytes")


if __name__ == "__main__":
    main()
