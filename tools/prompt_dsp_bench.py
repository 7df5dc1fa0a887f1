"""GPU box: the prompt-side DSP + codec `quantize` at the pipeline's real sizes -- a 15 s speaker prompt at 24 kHz (the longest the
pipeline keeps, infer_v2_5.py:627) -> resampling to 22.05 / 16 kHz, SeamlessM4T features, prompt log-mel, mean-normalised Kaldi fbank;
and `EnhancedCodec.quantize` of 749 w2v-bert frames with the shipped codec widths (8192 x 8 codebook, hidden 1024, Vocos 384 / 2048 x 12).
HIP-event times per call (median of 20 after 3 warm-ups), printed as one JSON line; run under rocprofv3 --kernel-trace --stats for the
per-kernel table (tools/gpu_call29.sh)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import audio as A, codec, synth  # noqa: E402

DEV = "cuda:0"


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def main():
    g = torch.Generator().manual_seed(3)
    sr, secs = 24000, 15
    wave = (torch.randn(1, sr * secs, generator=g) * 0.1).to(DEV)
    r22, r16 = A.Resample(sr, 22050, device=DEV), A.Resample(sr, 16000, device=DEV)
    fe = A.SeamlessM4TFeatureExtractor(device=DEV)
    a22, a16 = r22(wave), r16(wave)
    mel_kw = dict(n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024, fmin=0, fmax=None, center=False)
    out = {"prompt_seconds": secs, "source_rate": sr,
           "resample_24k_to_22k05_ms": timed(lambda: r22(wave)), "resample_24k_to_16k_ms": timed(lambda: r16(wave)),
           "seamless_features_ms": timed(lambda: fe(a16, sampling_rate=16000)),
           "mel_spectrogram_ms": timed(lambda: A.mel_spectrogram(a22, **mel_kw)),
           "kaldi_fbank_minus_mean_ms": timed(lambda: A.subtract_mean(A.fbank(a16, num_mel_bins=80, dither=0, sample_frequency=16000))),
           "frames": {"mel": int(A.mel_spectrogram(a22, **mel_kw).shape[2]), "fbank": int(A.fbank(a16, num_mel_bins=80).shape[0]),
                      "seamless": int(fe(a16, sampling_rate=16000)["input_features"].shape[1])}}
    c = codec.EnhancedCodec(**synth.CODEC_V2, device=DEV)
    sd = synth.codec_weights()
    sd.update(synth.codec_encoder_weights())
    c.load_state_dict(sd)
    x = torch.randn(1, out["frames"]["seamless"], 1024, generator=g).to(DEV)
    idx, q = c.quantize(x)
    out["codec_quantize_ms"] = timed(lambda: c.quantize(x), n=10)
    out["codec_quantize_rows"] = int(idx.shape[1])
    out["codec_quantize_distinct_codes"] = int(idx.unique().numel())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
