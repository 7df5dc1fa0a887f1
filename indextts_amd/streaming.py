"""Streaming synthesis: chunked GPT emission -> per-chunk codes-to-audio -> Hann cross-fade.

Mirrors `StreamingDecoder` of the reference's TensorRT pipeline (`backends/trt/pipeline/streaming.py:57-210`): the GPT engine
yields chunks of `chunk_size` codes that overlap by `overlap_size` codes (`UnifiedVoice.generate_chunks`, the decode loop stays
suspended on the device between chunks); every chunk goes through codes -> mel -> waveform on its own, the overlapping samples
of consecutive chunks are cross-faded with the two halves of a Hann window, and every row's last piece gets a short linear
fade-out.  Yields `(sample_rate, audio_list, done_list)` per chunk with int16 arrays (None for rows that finished earlier).

The flow-matching stage attends over a whole chunk, so a chunked utterance is NOT sample-identical to the one-shot synthesis
(the reference's streaming mode has the same property); where exactness matters the vocoder alone can be streamed exactly
(`BigVGAN.stream` / `open_stream`, overlap-save with the receptive-field halo).
"""
import time
from typing import Callable, List, Optional

import numpy as np

MEL_CODE_TO_FRAME_RATIO = 1.72        # mel frames per code (infer_v2.py:662)
HOP_SIZE = 256
SAMPLE_RATE = 22050
PCM16_MAX = 32767
TAIL_FADE_SAMPLES = 512


def overlap_samples(n_codes: int, frames_per_code: float = MEL_CODE_TO_FRAME_RATIO) -> int:
    """samples covered by `n_codes` codes (streaming.py:13-14; the reference's ratio is IndexTTS-2's 1.72 frames per code -- the v2.5
    pipeline renders int(2 * n * 1.72 * duration_factor) frames for n codes, infer_v2_5.py:833, and passes that ratio)"""
    return int(n_codes * frames_per_code) * HOP_SIZE


def crossfade(tail: np.ndarray, head: np.ndarray) -> np.ndarray:
    """`tail` fading out into `head` over their common length with a Hann window's falling / rising halves (streaming.py:17-27)."""
    n = min(len(tail), len(head))
    if n == 0:
        return np.zeros(0, dtype=np.float32)
    w = np.hanning(2 * n)
    return tail[:n] * w[n:] + head[:n] * w[:n]


def to_int16(audio: np.ndarray) -> np.ndarray:
    return np.clip(audio * PCM16_MAX, -PCM16_MAX, PCM16_MAX).astype(np.int16)


def fade_out_tail(audio: np.ndarray, fade_samples: int = TAIL_FADE_SAMPLES) -> np.ndarray:
    if len(audio) <= fade_samples:
        return audio
    out = audio.copy()
    out[-fade_samples:] *= np.linspace(1.0, 0.0, fade_samples)
    return out


class StreamingDecoder:
    """gpt_engine: an object with `generate_chunks(...)` yielding `(codes, is_last, batch_done, code_lens)` (UnifiedVoice);
    codes_to_audio_fn(codes, code_lens) -> list of float32 arrays in [-1, 1], one per row, covering that chunk's codes."""

    def __init__(self, gpt_engine, codes_to_audio_fn: Callable, chunk_size: int = 100, overlap_size: int = 20, verbose: bool = False,
                 frames_per_code: float = MEL_CODE_TO_FRAME_RATIO):
        if overlap_size >= chunk_size:
            raise ValueError(f"overlap_size ({overlap_size}) must be less than chunk_size ({chunk_size})")
        if not frames_per_code > 0:
            raise ValueError(f"frames_per_code must be positive, got {frames_per_code}")
        self.frames_per_code = float(frames_per_code)     # mel frames codes_to_audio_fn renders per code: sizes the cross-fade
        self.gpt_engine, self.codes_to_audio_fn = gpt_engine, codes_to_audio_fn
        self.chunk_size, self.overlap_size = int(chunk_size), int(overlap_size)
        self.stride = self.chunk_size - self.overlap_size
        self.verbose = verbose
        self.first_chunk_latency: Optional[float] = None

    def generate(self, inputs_embeds, attention_mask, max_new_tokens: int = 1500, **generation_kwargs):
        B = inputs_embeds.shape[0]
        ovlp = overlap_samples(self.overlap_size, self.frames_per_code)
        tails: List[Optional[np.ndarray]] = [None] * B      # a row's samples still waiting for the next chunk's head
        finished = [False] * B
        t0 = time.perf_counter()
        self.first_chunk_latency = None
        for idx, (codes, is_last, batch_done, code_lens) in enumerate(self.gpt_engine.generate_chunks(
                inputs_embeds, attention_mask, max_new_tokens, self.chunk_size, self.overlap_size, **generation_kwargs)):
            audios = self.codes_to_audio_fn(codes, code_lens)
            if self.first_chunk_latency is None:
                self.first_chunk_latency = time.perf_counter() - t0
            if self.verbose:
                print(f">> [streaming] chunk {idx}: {codes.shape[1]} codes -> {len(audios[0])} samples")
            out: List[Optional[np.ndarray]] = [None] * B
            done = [False] * B
            for b in range(B):
                if finished[b]:
                    continue
                audio = np.asarray(audios[b], dtype=np.float32)
                last_b = bool(batch_done[b]) or is_last
                if tails[b] is None:                         # the row's first chunk
                    piece, keep = (audio, None) if last_b else (audio[: len(audio) - ovlp], audio[len(audio) - ovlp:])
                else:
                    rest = audio[ovlp:]
                    blended = crossfade(tails[b], audio[:ovlp])
                    if last_b:
                        piece, keep = np.concatenate([blended, rest]), None
                    else:
                        piece, keep = np.concatenate([blended, rest[: len(rest) - ovlp]]), rest[len(rest) - ovlp:]
                if last_b:
                    out[b], done[b], finished[b], tails[b] = to_int16(fade_out_tail(piece)), True, True, None
                else:
                    out[b], tails[b] = to_int16(piece), keep
            yield SAMPLE_RATE, out, done
        if any(t is not None and not f for t, f in zip(tails, finished)):       # the engine stopped without a closing chunk
            out, done = [None] * B, [False] * B
            for b in range(B):
                if tails[b] is not None and not finished[b]:
                    out[b], done[b], finished[b] = to_int16(fade_out_tail(tails[b])), True, True
            yield SAMPLE_RATE, out, done
