"""CPU: oracle/audio_oracle.py (prompt-audio front end: resampling, log-mel of the 22 kHz prompt, Kaldi fbank, SeamlessM4T features)
against tests/golden/audio.npz = outputs of transformers' own SeamlessM4TFeatureExtractor and of the reference's own
`mel_spectrogram`, minted by tools/make_golden_audio.py; filter banks against transformers' implementations of the same published
definitions; the resampler (torchaudio absent: parity unpinned) against DSP properties."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import audio_oracle as AO
from tools.make_golden_audio import LENGTHS_16K, LENGTHS_22K


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "audio.npz"))


def test_filter_banks_and_window_vs_transformers():
    from transformers.audio_utils import mel_filter_bank, window_function
    tb = mel_filter_bank(num_frequency_bins=513, num_mel_filters=80, min_frequency=0.0, max_frequency=11025.0, sampling_rate=22050,
                         norm="slaney", mel_scale="slaney").T
    assert np.abs(AO.slaney_mel_basis(22050, 1024, 80, 0, None) - tb).max() <= 1e-8
    kb = mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20, max_frequency=8000, sampling_rate=16000, norm=None,
                         mel_scale="kaldi", triangularize_in_mel_space=True)
    assert np.abs(AO.kaldi_mel_banks().T - kb).max() <= 1e-12
    assert np.abs(AO.povey_window() - window_function(400, "povey", periodic=False)).max() <= 1e-12


def test_seamless_features_equal_transformers(gold):
    for i, n in enumerate(LENGTHS_16K):
        f, m = AO.seamless_features(gold[f"wave16k_{i}"])
        assert f.shape == gold[f"seamless_feat_{i}"].shape and np.array_equal(m, gold[f"seamless_mask_{i}"])
        assert np.abs(f - gold[f"seamless_feat_{i}"]).max() <= 1e-6
    assert int(gold["seamless_mask_1"].sum()) == gold["seamless_mask_1"].shape[1] - 1        # odd frame count: the padded pair is masked


def test_mel_spectrogram_equals_reference_function(gold):
    for i, n in enumerate(LENGTHS_22K):
        m = AO.mel_spectrogram(torch.from_numpy(gold[f"wave22k_{i}"])[None])[0].numpy()
        assert m.shape == gold[f"refmel_{i}"].shape == (80, 1 + (n + 768 - 1024) // 256)
        assert np.abs(m - gold[f"refmel_{i}"]).max() <= 1e-6


def test_kaldi_fbank_tracks_kaldi_compatible_anchor(gold):
    for i, n in enumerate(LENGTHS_16K):
        x = torch.from_numpy(gold[f"wave16k_{i}"])[None]
        anchor = gold[f"kaldi_anchor_{i}"]
        k64 = AO.kaldi_fbank(x, dtype=torch.float64).numpy()
        k32 = AO.kaldi_fbank(x).numpy()
        assert k32.shape == anchor.shape == (1 + (n - 400) // 160, 80)
        assert np.abs(k64 - anchor).max() <= 5e-6                       # the anchor rounds its spectrum to complex64
        # float32 (torchaudio's arithmetic): FFT rounding shows in bins 80 dB below the frame's peak only
        loud = anchor > anchor.max() - 12.0
        assert np.abs(k32 - anchor)[loud].max() <= 2e-4 and np.abs(k32 - anchor).max() <= 1e-2
    assert AO.kaldi_fbank(torch.zeros(1, 399)).shape == (0, 80)


def test_resample_properties():
    for orig, new in ((24000, 22050), (24000, 16000), (44100, 16000), (16000, 22050)):
        n = 6000
        f = 440.0
        x = torch.sin(2 * math.pi * f * torch.arange(n, dtype=torch.float64) / orig).float()
        y = AO.resample(x[None], orig, new)
        assert y.shape == (1, math.ceil(new * n / orig))
        t = torch.arange(y.shape[1], dtype=torch.float64) / new
        ref = torch.sin(2 * math.pi * f * t).float()
        k, width, o, _ = AO.sinc_resample_kernel(orig, new)
        edge = int(2 * width * new / orig) + 2                           # the zero padding reaches this far into the output
        err = (y[0] - ref)[edge:-edge].abs().max()
        assert float(err) <= 2e-2, (orig, new, float(err))              # 0.99 roll-off: a 440 Hz tone passes with ~1 % gain error
        assert abs(float(k.sum(1).mean()) - 1.0) <= 2e-2                # unit DC gain per phase
    x = torch.randn(2, 777)
    assert AO.resample(x, 16000, 16000) is x
    assert AO.resample(x, 48000, 16000).shape == (2, 259)
