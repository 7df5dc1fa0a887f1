"""GPU parity of the prompt-audio front end (indextts_amd/audio.py -> audio_kernels.hip through the C ABI; SURVEY.md section 8 f-3, the
DSP half) against tests/golden/audio.npz = outputs of transformers' own SeamlessM4TFeatureExtractor and of the reference's own
`mel_spectrogram` (tools/make_golden_audio.py), and against the oracle for the two torchaudio pieces (Kaldi fbank, resampler).

Tolerances (float32 kernel vs float64 / float32 CPU references; the float32 numpy model of the kernel in tests/test_host_audio.py lands
at 1.5e-4 / 8e-6 on the same data):
    SeamlessM4T features   5e-3 absolute on per-bin z-scores (the reference computes in float64; bins ~80 dB below a frame's peak carry
                           the float32 FFT's rounding)
    log-mel of the prompt  2e-4 absolute on values in -11.5 .. 1.1
    Kaldi fbank            2e-4 on bins within 12 nepers of the maximum, 1e-2 elsewhere (same rule as the float32 oracle vs the anchor)
    resampler              1e-5 absolute on signals of amplitude <= 1 (same taps, float32 accumulation)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import audio_oracle as AO
from tools.make_golden_audio import LENGTHS_16K, LENGTHS_22K

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "audio.npz"))


def test_seamless_features_vs_transformers(gold):
    from indextts_amd.audio import SeamlessM4TFeatureExtractor
    fe = SeamlessM4TFeatureExtractor.from_pretrained("unused", device=DEV)
    for i, n in enumerate(LENGTHS_16K):
        x = torch.from_numpy(gold[f"wave16k_{i}"])
        o = fe(x[None], sampling_rate=16000, return_tensors="pt")                      # (1, L) tensor: how the pipeline calls it
        f, m = o["input_features"].cpu().numpy(), o["attention_mask"].cpu().numpy()
        err = np.abs(f - gold[f"seamless_feat_{i}"]).max()
        print(f"SeamlessM4T features n={n}: {f.shape}, max|d| vs transformers {err:.2e}")
        assert f.shape == gold[f"seamless_feat_{i}"].shape and err <= 5e-3
        assert m.dtype == np.int32 and np.array_equal(m, gold[f"seamless_mask_{i}"])
    # a batch of two: right-padded to the longer one, each row normalised over its own frames
    o = fe([gold["wave16k_0"], gold["wave16k_1"]], sampling_rate=16000)
    f, m = o["input_features"].cpu().numpy(), o["attention_mask"].cpu().numpy()
    n0, n1 = gold["seamless_feat_0"].shape[1], gold["seamless_feat_1"].shape[1]
    assert f.shape == (2, n1, 160) and m[0].sum() == n0 and m[1].sum() == n1 - 1
    assert np.abs(f[0, :n0] - gold["seamless_feat_0"][0]).max() <= 5e-3 and np.abs(f[0, n0:]).max() == 0.0
    assert np.abs(f[1] - gold["seamless_feat_1"][0]).max() <= 5e-3
    with pytest.raises(ValueError):
        fe(x, sampling_rate=8000)


def test_mel_spectrogram_vs_reference_function(gold):
    from indextts_amd.audio import mel_spectrogram
    kw = dict(n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024, fmin=0, fmax=None, center=False)
    for i, n in enumerate(LENGTHS_22K):
        x = torch.from_numpy(gold[f"wave22k_{i}"])[None].to(DEV)
        m = mel_spectrogram(x, **kw)
        err = float((m[0].cpu() - torch.from_numpy(gold[f"refmel_{i}"])).abs().max())
        print(f"mel_spectrogram n={n}: {tuple(m.shape)}, max|d| vs the reference function {err:.2e}")
        assert m.shape == (1, 80, gold[f"refmel_{i}"].shape[1]) and err <= 2e-4
    # batch rows are independent; fmax = 8000 (the other branch of infer_v2_5.py:262) against the oracle
    a, b = gold["wave22k_0"], gold["wave22k_1"][: gold["wave22k_0"].size]
    both = mel_spectrogram(torch.from_numpy(np.stack([a, b])).to(DEV), **kw).cpu()
    assert float((both[0] - torch.from_numpy(gold["refmel_0"])).abs().max()) <= 2e-4
    assert float((both[1] - AO.mel_spectrogram(torch.from_numpy(b)[None])[0]).abs().max()) <= 2e-4
    kw8 = dict(kw, fmax=8000)
    m8 = mel_spectrogram(torch.from_numpy(a)[None].to(DEV), **kw8).cpu()
    assert float((m8 - AO.mel_spectrogram(torch.from_numpy(a)[None], fmax=8000)).abs().max()) <= 2e-4
    with pytest.raises(ValueError):
        mel_spectrogram(torch.zeros(1, 300, device=DEV), **kw)                 # shorter than the reflect padding: torch's pad raises too


def test_kaldi_fbank_vs_oracle_and_anchor(gold):
    from indextts_amd.audio import fbank, subtract_mean
    for i, n in enumerate(LENGTHS_16K):
        x = torch.from_numpy(gold[f"wave16k_{i}"])[None]
        f = fbank(x.to(DEV), num_mel_bins=80, dither=0, sample_frequency=16000)
        anchor = gold[f"kaldi_anchor_{i}"]
        assert f.shape == anchor.shape
        d = np.abs(f.cpu().numpy() - anchor)
        loud = anchor > anchor.max() - 12.0
        d32 = np.abs(f.cpu().numpy() - AO.kaldi_fbank(x).numpy())
        print(f"kaldi fbank n={n}: max|d| vs the float64 anchor {d.max():.2e} (loud bins {d[loud].max():.2e}); vs the float32 oracle {d32.max():.2e}")
        assert d[loud].max() <= 2e-4 and d.max() <= 1e-2
        c = subtract_mean(f)
        assert float((c.cpu() - (f.cpu() - f.cpu().mean(dim=0, keepdim=True))).abs().max()) <= 1e-5
    assert fbank(torch.zeros(1, 399, device=DEV), num_mel_bins=80).shape == (0, 80)         # shorter than one frame
    one = fbank(torch.from_numpy(gold["wave16k_0"][:400])[None].to(DEV), num_mel_bins=80)
    assert one.shape == (1, 80) and float((one.cpu() - AO.kaldi_fbank(torch.from_numpy(gold["wave16k_0"][:400])[None])).abs().max()) <= 1e-2


def test_resample_vs_oracle(gold):
    from indextts_amd.audio import Resample
    x = torch.from_numpy(gold["wave22k_1"])
    for orig, new in ((22050, 16000), (24000, 22050), (44100, 16000), (16000, 22050), (48000, 16000)):
        y = Resample(orig, new, device=DEV)(x[None])
        ref = AO.resample(x[None], orig, new)
        err = float((y.cpu() - ref).abs().max())
        print(f"resample {orig} -> {new}: {tuple(y.shape)}, max|d| vs the oracle {err:.2e}")
        assert y.shape == ref.shape == (1, math.ceil(new * x.numel() / orig)) and err <= 1e-5
    two = torch.stack([x[:5000], x[5000:10000]])
    y = Resample(24000, 16000, device=DEV)(two.to(DEV))
    assert float((y.cpu() - AO.resample(two, 24000, 16000)).abs().max()) <= 1e-5
    assert Resample(16000, 16000, device=DEV)(x) is x


def test_colnorm_unit():
    from indextts_amd.audio import _colnorm
    g = torch.Generator().manual_seed(4)
    x = torch.randn(333, 80, generator=g) * 3 + 5
    out = torch.zeros(334, 80)
    z = _colnorm(x.to(DEV), out.to(DEV), mode=1, ddof=1, eps=1e-7).cpu()
    ref = (x - x.mean(0, keepdim=True)) / torch.sqrt(x.var(0, unbiased=True, keepdim=True) + 1e-7)
    assert float((z[:333] - ref).abs().max()) <= 1e-5 and float(z[333].abs().max()) == 0.0


def test_reference_frontend_runs_the_dsp_on_the_engine(gold):
    """ReferenceFrontend.speaker_bundle with a stand-in reference object (tests/test_reference_frontend.py): resampling, SeamlessM4T
    features, prompt log-mel and the mean-normalised Kaldi fbank come from indextts_amd/audio.py; what the stand-in's encoders receive is
    compared with the oracle chain on the same waveform."""
    from indextts_amd.infer_v2_5 import ReferenceFrontend
    from tests.test_reference_frontend import FakeRefIndexTTS2
    wave = torch.from_numpy(gold["wave22k_1"])[None]
    seen = {}

    class Ref(FakeRefIndexTTS2):
        def _load_and_cut_audio(self, path, seconds, verbose=False, sr=None):
            return wave, 24000

        def get_emb(self, feats, mask):
            seen["feats"], seen["mask"] = feats.cpu(), mask.cpu()
            return torch.zeros(1, feats.shape[1], 1024, device=DEV)

    ref = Ref()
    ref.extract_features = None                                                   # must not be used
    ref.mel_fn = None

    def campplus(feat):
        seen["fbank"] = feat.cpu()
        return torch.ones(1, 192, device=DEV)

    ref.campplus_model = campplus
    cfg = {"s2mel": {"preprocess_params": {"sr": 22050, "spect_params": {"n_fft": 1024, "win_length": 1024, "hop_length": 256, "n_mels": 80,
                                                                         "fmin": 0, "fmax": "None"}}}}
    fe = ReferenceFrontend(cfg, "unused", DEV, ref=ref)
    assert fe.audio is not None
    b = fe.speaker_bundle("prompt.wav")
    a16, a22 = AO.resample(wave, 24000, 16000), AO.resample(wave, 24000, 22050)
    mel = AO.mel_spectrogram(a22)
    assert b["ref_mel"].shape == mel.shape and float((b["ref_mel"].cpu() - mel).abs().max()) <= 5e-4
    f, m = AO.seamless_features(a16[0].numpy())
    assert seen["feats"].shape == f.shape and float((seen["feats"] - torch.from_numpy(f)).abs().max()) <= 5e-3
    assert np.array_equal(seen["mask"].numpy(), m)
    k = AO.kaldi_fbank(a16)
    k = k - k.mean(dim=0, keepdim=True)
    assert seen["fbank"].shape == (1,) + tuple(k.shape) and float((seen["fbank"][0] - k).abs().max()) <= 1e-2
    assert b["prompt_condition"].shape == (1, mel.shape[2], 512) and b["style"].shape == (1, 192)


def test_full_length_prompt_chain_vs_oracle():
    """The longest prompt the pipeline keeps (15 s, infer_v2_5.py:627) at 24 kHz through the whole DSP chain, engine vs oracle: resample to
    16 / 22.05 kHz, SeamlessM4T features (749 stacked frames), prompt log-mel (1291 frames), mean-normalised Kaldi fbank (1498 frames)."""
    from indextts_amd import audio as A
    from tools.make_golden_audio import speechlike
    x = torch.from_numpy(speechlike(15 * 24000, 24000, 77))[None]
    a16, a22 = A.Resample(24000, 16000, device=DEV)(x), A.Resample(24000, 22050, device=DEV)(x)
    o16, o22 = AO.resample(x, 24000, 16000), AO.resample(x, 24000, 22050)
    assert a16.shape == o16.shape == (1, 240000) and a22.shape == o22.shape == (1, 330750)
    assert float((a16.cpu() - o16).abs().max()) <= 1e-5 and float((a22.cpu() - o22).abs().max()) <= 1e-5
    f = A.SeamlessM4TFeatureExtractor(device=DEV)(a16, sampling_rate=16000)
    fo, mo = AO.seamless_features(o16[0].numpy())
    err_f = float((f["input_features"].cpu() - torch.from_numpy(fo)).abs().max())
    assert f["input_features"].shape == fo.shape == (1, 749, 160) and np.array_equal(f["attention_mask"].cpu().numpy(), mo) and err_f <= 5e-3
    mel = A.mel_spectrogram(a22, n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024, fmin=0, fmax=None, center=False)
    err_m = float((mel.cpu() - AO.mel_spectrogram(o22)).abs().max())
    assert mel.shape == (1, 80, 1291) and err_m <= 2e-4
    k = A.subtract_mean(A.fbank(a16, num_mel_bins=80, dither=0, sample_frequency=16000)).cpu()
    ko = AO.kaldi_fbank(o16, dtype=torch.float64)
    ko = (ko - ko.mean(dim=0, keepdim=True)).float()
    loud = ko > ko.max() - 12.0
    err_k = (k - ko).abs()
    print(f"15 s prompt: SeamlessM4T {err_f:.2e}, log-mel {err_m:.2e}, fbank {float(err_k.max()):.2e} (loud bins {float(err_k[loud].max()):.2e})")
    assert k.shape == (1498, 80) and float(err_k[loud].max()) <= 2e-4 and float(err_k.max()) <= 1e-2


def test_v1_conditioning_mel_vs_oracle(gold):
    """`MelSpectrogramFeatures` (IndexTTS-1 / 1.5 conditioning mel, indextts/utils/feature_extractors.py:24-51): 24 kHz, centred reflect padding,
    100 HTK bins on the magnitude spectrum, log clip 1e-7 -- engine vs the torch.stft restatement (torchaudio absent: parity unpinned beyond
    the bank's agreement with transformers' HTK bank, tests/test_host_audio.py)."""
    from indextts_amd.audio import MelSpectrogramFeatures
    x = torch.from_numpy(np.stack([gold["wave22k_1"][:26000], gold["wave22k_1"][4000:30000]]))
    for padding in ("center", "same"):
        m = MelSpectrogramFeatures(padding=padding, device=DEV)(x)
        ref = AO.mel_spectrogram_features(x, padding=padding)
        err = float((m.cpu() - ref).abs().max())
        print(f"v1 conditioning mel ({padding}): {tuple(m.shape)}, max|d| vs the oracle {err:.2e} (range {float(ref.min()):.1f}..{float(ref.max()):.1f})")
        assert m.shape == ref.shape and err <= 5e-4
    with pytest.raises(NotImplementedError):
        MelSpectrogramFeatures(normalize=True)
