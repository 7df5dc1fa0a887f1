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


# ---- second, independent cross-checks of the two torchaudio pieces that cannot be pinned to torchaudio itself -------
def _hann_sinc(t, base, width=6):
    """torchaudio's published interpolation kernel as a function of continuous time t (in input samples / orig-rate units):
    base * sinc(base t) * cos^2(pi base t / (2 width)) for |base t| < width, 0 outside (functional.py `_get_sinc_resample_kernel`,
    resampling_method="sinc_interp_hann")."""
    bt = np.clip(np.asarray(t, dtype=np.float64) * base, -width, width)
    return base * np.sinc(bt) * np.cos(bt * np.pi / width / 2) ** 2


@pytest.mark.parametrize("orig,new", [(24000, 22050), (24000, 16000), (44100, 16000), (16000, 22050), (22050, 16000)])
def test_resample_equals_independent_polyphase_implementation(orig, new):
    """The restated resampler (frame the padded input, one strided conv1d per output phase) against scipy's polyphase resampler
    (`resample_poly`: zero-stuff by up, ONE prototype FIR at the common rate, keep every down-th sample) fed with the SAME published
    kernel sampled on the common-rate grid.  The two share no code: this checks the phase / stride / padding / length logic of the
    restatement, the sine test above checks the kernel formula against the analytic answer."""
    from scipy.signal import resample_poly
    g = math.gcd(orig, new)
    o, n = orig // g, new // g
    base = min(o, n) * 0.99
    width = math.ceil(6 * o / base)
    half = (width + 1) * n                                               # prototype taps on the grid of rate orig * n / g, centred
    m = np.arange(-half, half + 1, dtype=np.float64)
    proto = _hann_sinc(m / n, base / o) / 1.0                            # t in input samples = m / n
    x = np.random.RandomState(7).randn(3000).astype(np.float32)
    # resample_poly multiplies the taps by `up` (unit-gain convention of a zero-stuffed signal); the published kernel already carries
    # its gain (base / o per input sample), so divide it out
    want = resample_poly(x.astype(np.float64), n, o, window=proto / n, padtype="constant")
    got = AO.resample(torch.from_numpy(x)[None], orig, new)[0].double().numpy()
    assert got.shape == want.shape == (math.ceil(n * x.size / o),)
    err = np.abs(got - want).max()
    assert err <= 5e-6 * max(1.0, np.abs(want).max()), (orig, new, err)


def test_resample_bandlimited_signal_matches_its_analytic_resampling():
    """A band-limited multi-tone (all components below 0.4 x the lower Nyquist) resampled 24 kHz -> 16 kHz and 16 -> 22.05 kHz equals
    the same multi-tone evaluated on the new grid to the kernel's pass-band ripple."""
    rs = np.random.RandomState(3)
    for orig, new in ((24000, 16000), (16000, 22050)):
        f = rs.uniform(60.0, 0.4 * min(orig, new) / 2, 12)
        a, ph = rs.uniform(0.02, 0.08, 12), rs.uniform(0, 2 * np.pi, 12)
        sig = lambda t: sum(ai * np.sin(2 * np.pi * fi * t + pi) for ai, fi, pi in zip(a, f, ph))
        nsamp = 9000
        x = torch.from_numpy(sig(np.arange(nsamp) / orig).astype(np.float32))[None]
        y = AO.resample(x, orig, new)[0].double().numpy()
        ref = sig(np.arange(y.size) / new)
        edge = 200
        assert np.abs(y - ref)[edge:-edge].max() <= 0.02 * np.abs(ref).max()


def test_kaldi_fbank_float32_vs_transformers_on_a_15s_prompt():
    """CAMPPlus' exact input (infer_v2_5.py:644-647: 16 kHz, 80 bins, dither 0, then per-bin mean removal) on a 15 s prompt -- the
    longest the pipeline admits -- in float32, against transformers' Kaldi-compatible `spectrogram` (an independent implementation of the
    published Kaldi recipe) and, downstream, what the difference does to the CAMPPlus style vector."""
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    from tools.make_golden_audio import speechlike
    from oracle import campplus_oracle as CO
    x = speechlike(15 * 16000, 16000, 91)
    kb = mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20, max_frequency=8000, sampling_rate=16000, norm=None,
                         mel_scale="kaldi", triangularize_in_mel_space=True)
    anchor = spectrogram(x.astype(np.float64), window_function(400, "povey", periodic=False), frame_length=400, hop_length=160,
                         fft_length=512, power=2.0, center=False, preemphasis=0.97, mel_filters=kb, log_mel="log", mel_floor=AO.EPS_F32,
                         remove_dc_offset=True).T
    k32 = AO.kaldi_fbank(torch.from_numpy(x)[None]).numpy()
    assert k32.shape == anchor.shape == (1498, 80)
    loud = anchor > anchor.max() - 12.0
    d = np.abs(k32 - anchor)
    assert d[loud].max() <= 2e-4 and d.max() <= 1e-2
    # downstream effect on the speaker embedding: style vector from either fbank through the CAMPPlus oracle
    sd = CO.synth_weights()
    with torch.no_grad():
        feat = lambda a: torch.from_numpy((a - a.mean(0, keepdims=True)).astype(np.float32))[None]
        s_a, s_b = CO.campplus(sd, feat(k32)), CO.campplus(sd, feat(anchor.astype(np.float32)))
    cos = float(torch.nn.functional.cosine_similarity(s_a, s_b).item())
    rel = float((s_a - s_b).norm() / s_b.norm())
    print(f"CAMPPlus style from the float32 fbank vs from transformers' Kaldi-compatible features: cosine {cos:.8f}, relative L2 {rel:.2e}")
    assert cos >= 1.0 - 1e-5 and rel <= 2e-3
