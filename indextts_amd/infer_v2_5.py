"""`IndexTTS2` (v2.5) pipeline class with the reference's constructor / `infer()` / `infer_generator()` signatures,
with the two hot stages re-routed to the HIP engine.

Reference mirrored: `indextts/infer_v2_5.py::IndexTTS2` (`__init__` :77-279, `infer` :506-567, `infer_generator`
:570-899).  What changes behind the API:
  * `self.gpt` is `indextts_amd.gpt.UnifiedVoice` (decode on the HIP engine), `self.bigvgan` is
    `indextts_amd.bigvgan.BigVGAN`;
  * all text segments of one call are decoded as ONE left-padded GPT batch and vocoded as ONE ragged BigVGAN batch
    (the reference loops segment by segment, `:749`); per-segment results are unchanged (padding invariance is a tested
    property), and `infer_batch()` extends the same to many utterances;
  * codes -> mel (semantic-codec decode, length regulator, 25-step CFG flow matching) runs on the HIP engine too when its stages are
    given or buildable from the frontend's state dicts (`codes_to_mel=` "auto" / "engine" / "frontend", recorded in
    `self.codes_to_mel_mode`); "frontend" keeps that stage on the frontend's PyTorch modules;
  * what is NOT on the engine -- audio file decoding, text normalisation + tokenizer + segment splitting, the QwenEmotion LLM -- is
    reached through a `frontend` object.  `ReferenceFrontend` builds the reference's own modules from the reference package +
    checkpoint directory (needs the reference's `indextts` package, torchaudio, librosa, ... -- none exist in the build/bench images,
    so that class is exercised only where they do) and moves the prompt encoders / DSP onto the engine where their configuration
    is the shipped one; `indextts_amd.frontend.EngineFrontend*` need no reference package; tests inject a stub with the same methods.
There is no silent fallback: a missing frontend dependency raises at construction.
"""
import os
import time
import warnings
import wave
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

PCM16_MAX = 32767.0


def save_pcm_wav(path: str, wav: torch.Tensor, sampling_rate: int):
    """16-bit PCM WAV writer for a PCM-scale (+-32767) waveform (C, T) -- `indextts/utils/common.py:38-58` semantics
    (normalise by 32767, clamp to [-1, 1], encode PCM_S16) without the torchaudio dependency."""
    w = (wav.detach().to("cpu", torch.float32) / PCM16_MAX).clamp_(-1.0, 1.0)
    pcm = torch.round(w * PCM16_MAX).to(torch.int16).numpy()
    if pcm.ndim == 1:
        pcm = pcm[None, :]
    with wave.open(path, "wb") as f:
        f.setnchannels(pcm.shape[0])
        f.setsampwidth(2)
        f.setframerate(int(sampling_rate))
        f.writeframes(np.ascontiguousarray(pcm.T).tobytes())


class Frontend:
    """Everything outside the hot path, as the pipeline consumes it.  Shapes follow the reference."""

    def speaker_bundle(self, spk_audio_prompt) -> Dict[str, torch.Tensor]:
        """-> {style (1,192), spk_cond_emb (1,T,1024), ref_mel (1,80,Tm), prompt_condition (1,Tm,512)}  (:620-667)"""
        raise NotImplementedError

    def emo_cond(self, emo_audio_prompt) -> torch.Tensor:            # (:682-697)
        raise NotImplementedError

    def merge_emovec(self, spk_cond_emb, emo_cond_emb, alpha: float) -> torch.Tensor:   # gpt.merge_emovec (:759-765)
        raise NotImplementedError

    def emo_vector_mix(self, emo_vector, style, use_random: bool):  # (:669-680) -> (emovec_mat (1,D), weight_sum)
        raise NotImplementedError

    def text_segments(self, text: str, lang: str, max_text_tokens_per_segment: int, text_normalization: bool,
                      capacity: int) -> List[torch.Tensor]:
        """normalise, split (split_text_by_tokens :427-464), tokenize `<|lang|> seg`, append the stop id (:699-727)."""
        raise NotImplementedError

    def lang_id(self, lang: str) -> int:
        raise NotImplementedError

    def codes_to_mel(self, codes: torch.Tensor, code_lens: torch.Tensor, bundle, duration_factor: float):
        """semantic_codec.decode -> length_regulator -> cfm.inference -> drop prompt frames (:830-846).
        -> mel (B, 80, Tmax) f32 and lens (B,) frames."""
        raise NotImplementedError


class IndexTTS2:
    USE_GPT_LATENT = False                 # the s2mel stage of IndexTTS-2 carries the GPT-latent projector `gpt_layer` (subclass sets it)
    SPK_COND_MODE = "campplus"             # `UnifiedVoice(..., spk_cond_mode=)` of this version (infer_v2_5.py:106; v2: the default, conformer)

    @staticmethod
    def _bigvgan_dir(model_dir, aux_paths=None):
        """infer_v2_5.py:224-228: `<model_dir>/hf_cache/bigvgan`, else the pre-downloaded `aux_paths["bigvgan"]`."""
        d = os.path.join(model_dir, "hf_cache", "bigvgan")
        if not os.path.isdir(d) and aux_paths and "bigvgan" in aux_paths:
            d = aux_paths["bigvgan"]
        return d

    def __init__(self, cfg_path="checkpoints/config.yaml", model_dir="checkpoints", use_bf16=False, device=None,
                 use_cuda_kernel=None, use_deepspeed=False, use_accel=False, use_torch_compile=False, use_qwen_emo=False,
                 *, frontend: Optional[Frontend] = None, gpt=None, bigvgan=None, cfg: Optional[dict] = None,
                 semantic_codec=None, s2mel=None, aux_paths=None, codes_to_mel="auto"):
        """`codes_to_mel`: where codes -> mel runs.  "engine": the HIP codec / length regulator / CFM stages (given as
        `semantic_codec=` / `s2mel=` or built from the frontend's state dicts) -- raises if they cannot be built; "frontend": the
        frontend's PyTorch `codes_to_mel` (north_star keeps s2mel on PyTorch as a legitimate integration mode); "auto" (default):
        the engine when its stages are available, else the frontend, and the choice is recorded in `self.codes_to_mel_mode`."""
        if codes_to_mel not in ("auto", "engine", "frontend"):
            raise ValueError(f"codes_to_mel must be 'auto', 'engine' or 'frontend', got {codes_to_mel!r}")
        if device is not None:
            self.device = device
        elif torch.cuda.is_available():
            self.device = "cuda:0"
        else:
            raise RuntimeError("IndexTTS2 (HIP engine) needs a GPU: there is no CPU path")
        self.use_bf16 = bool(use_bf16)
        self.use_cuda_kernel = True            # the HIP kernels are the only implementation
        self.model_dir = model_dir
        self.dtype = torch.bfloat16 if self.use_bf16 else None
        if cfg is None:
            import yaml
            with open(cfg_path) as f:
                cfg = yaml.safe_load(f)
        self.cfg = cfg
        gcfg = dict(cfg["gpt"])
        self.stop_mel_token = gcfg.get("stop_mel_token", 8193)
        self.model_version = cfg.get("version", None)
        self.gr_progress = None
        self.low_vram = False                  # infer_v2_5.py:124-131 turns it on below 10 GB of device memory; callers may set it
        self.qwen_emo = None
        if use_qwen_emo:
            raise NotImplementedError("QwenEmotion (text -> emotion vector) is a prompt-side LLM; run it with the reference "
                                      "package and pass emo_vector= instead")
        if gpt is None:
            from .gpt import UnifiedVoice
            gcfg.pop("spk_cond_mode", None)
            gpt = UnifiedVoice(**gcfg, spk_cond_mode=self.SPK_COND_MODE, precision="bf16" if self.use_bf16 else "fp32",
                               device=self.device)
            ck = torch.load(os.path.join(model_dir, cfg["gpt_checkpoint"]), map_location="cpu")
            gpt.load_state_dict(ck["model"] if "model" in ck else ck)
            gpt.post_init_gpt2_config(use_deepspeed=False, kv_cache=True, half=self.use_bf16)
        self.gpt = gpt
        if bigvgan is None:
            from .bigvgan import BigVGAN
            bigvgan = BigVGAN.from_pretrained(self._bigvgan_dir(model_dir, aux_paths)).to(self.device)
        self.bigvgan = bigvgan
        # codes -> mel on the HIP engine when both stages are given (indextts_amd.codec.EnhancedCodec, indextts_amd.s2mel.MyModel);
        # otherwise the frontend's codes_to_mel (the reference's PyTorch modules) is used
        self.semantic_codec = semantic_codec
        self.s2mel = s2mel
        if frontend is None:
            frontend = ReferenceFrontend(cfg, model_dir, self.device, self.gpt, cfg_path=cfg_path)
        self.frontend = frontend
        if codes_to_mel == "frontend":
            self.semantic_codec = self.s2mel = None
        elif (self.s2mel is None or self.semantic_codec is None) and hasattr(frontend, "engine_state_dicts") and "s2mel" in cfg \
                and "semantic_codec" in cfg:
            # codes -> mel on the engine, weights from the state dicts the frontend read (the reference's modules, or codec.pth / s2mel.pth
            # themselves); a stage that was injected is kept
            from .codec import EnhancedCodec
            from .s2mel import MyModel
            sds = frontend.engine_state_dicts()
            if self.semantic_codec is None:
                self.semantic_codec = EnhancedCodec(**dict(cfg["semantic_codec"]), device=self.device)
                self.semantic_codec.load_state_dict(sds["semantic_codec"])
            if self.s2mel is None:
                net = {"cfm": sds["cfm"], "length_regulator": sds["length_regulator"]}
                if self.USE_GPT_LATENT:                   # IndexTTS-2: MyModel(cfg.s2mel, use_gpt_latent=True), infer_v2.py:145
                    if "gpt_layer" not in sds:
                        raise RuntimeError("IndexTTS-2 needs the s2mel checkpoint's `gpt_layer` (latent projector): this frontend's "
                                           "engine_state_dicts() has none -- use indextts_amd.frontend.EngineFrontendV2 or inject s2mel=")
                    net["gpt_layer"] = sds["gpt_layer"]
                self.s2mel = MyModel(cfg["s2mel"], use_gpt_latent=self.USE_GPT_LATENT, precision="bf16" if self.use_bf16 else "fp32",
                                     device=self.device)
                self.s2mel.load_state_dict(net)
        if codes_to_mel == "engine" and (self.s2mel is None or self.semantic_codec is None):
            raise RuntimeError("codes_to_mel='engine': the HIP codec / s2mel stages were neither injected (semantic_codec=, s2mel=) nor "
                               "buildable from this frontend (it needs engine_state_dicts() and cfg['s2mel'] / cfg['semantic_codec'])")
        self.codes_to_mel_mode = "engine" if (self.s2mel is not None and self.semantic_codec is not None) else "frontend"
        self.tokenizer = getattr(frontend, "tokenizer", None)
        # reference cache attributes (:269-279)
        self.cache_spk_cond = None
        self.cache_s2mel_style = None
        self.cache_s2mel_prompt = None
        self.cache_spk_audio_prompt = None
        self.cache_emo_cond = None
        self.cache_emo_audio_prompt = None
        self.cache_mel = None
        self._bundle = None
        self.last_timing = {}

    # ---- helpers mirrored from the reference ---------------------------------------------------------------------
    def _set_gr_progress(self, value, desc):
        if self.gr_progress is not None:
            self.gr_progress(value, desc=desc)

    def interval_silence(self, wavs, sampling_rate=22050, interval_silence=200):
        if not wavs or interval_silence <= 0:
            return wavs
        return torch.zeros(wavs[0].size(0), int(sampling_rate * interval_silence / 1000.0))

    def insert_interval_silence(self, wavs, sampling_rate=22050, interval_silence=200):
        if not wavs or interval_silence <= 0:
            return wavs
        sil = torch.zeros(wavs[0].size(0), int(sampling_rate * interval_silence / 1000.0))
        out = []
        for i, w in enumerate(wavs):
            out.append(w)
            if i < len(wavs) - 1:
                out.append(sil)
        return out

    def trim_codes(self, codes: torch.Tensor):
        """Stop-token trim (:809-821): per row the length up to the first stop token, batch cut to the longest."""
        lens = []
        for code in codes:
            hit = (code == self.stop_mel_token).nonzero(as_tuple=False)
            lens.append(int(hit[0, 0]) if hit.numel() > 0 else int(code.numel()))
        mx = max(lens) if lens else 0
        return codes[:, :mx], torch.tensor(lens, dtype=torch.long, device=codes.device)

    # ---- conditioning --------------------------------------------------------------------------------------------
    def _speaker(self, spk_audio_prompt):
        if self._bundle is None or self.cache_spk_audio_prompt != spk_audio_prompt:
            self._bundle = self.frontend.speaker_bundle(spk_audio_prompt)
            self.cache_spk_audio_prompt = spk_audio_prompt
            self.cache_spk_cond = self._bundle["spk_cond_emb"]
            self.cache_s2mel_style = self._bundle["style"]
            self.cache_s2mel_prompt = self._bundle["prompt_condition"]
            self.cache_mel = self._bundle["ref_mel"]
        return self._bundle

    def _emotion(self, emo_audio_prompt):
        if self.cache_emo_cond is None or self.cache_emo_audio_prompt != emo_audio_prompt:
            self.cache_emo_cond = self.frontend.emo_cond(emo_audio_prompt)
            self.cache_emo_audio_prompt = emo_audio_prompt
        return self.cache_emo_cond

    def _emovec(self, bundle, emo_audio_prompt, emo_alpha, emo_vector, use_random):
        emo_cond_emb = self._emotion(emo_audio_prompt)
        if getattr(self.gpt, "cond_encoders", None) is not None:      # emotion Conformer + Perceiver on the engine (indextts_amd/cond.py)
            spk = bundle["spk_cond_emb"]
            # the reference passes the feature WIDTH (1024) as the "length" (:760-765): every frame of a
            # prompt shorter than 1024 frames is valid, a longer one is cut there
            emovec = self.gpt.merge_emovec(spk, emo_cond_emb, torch.tensor([min(spk.shape[-1], spk.shape[1])]),
                                           torch.tensor([min(emo_cond_emb.shape[-1], emo_cond_emb.shape[1])]), alpha=emo_alpha)
        else:
            emovec = self.frontend.merge_emovec(bundle["spk_cond_emb"], emo_cond_emb, emo_alpha)
        if emo_vector is not None:
            emovec_mat, wsum = self.frontend.emo_vector_mix(emo_vector, bundle["style"], use_random)
            emovec = emovec_mat + (1 - wsum) * emovec                 # (:767-769)
        return emovec

    # ---- API -------------------------------------------------------------------------------------------------------
    @staticmethod
    def split_text_by_punctuation(text, max_chars=40):
        """infer_v2_5.py:466-487: pieces of at most `max_chars` characters cut after punctuation; a piece with no punctuation
        stays whole (no mid-word cut)."""
        import re
        pieces, cur = [], ""
        for part in re.split(r'(?<=[\uff0c\u3002\uff01\uff1f\u3001\uff1b\uff1a,\.!\?;:\n])', text):
            if not part:
                continue
            if len(cur) + len(part) <= max_chars:
                cur += part
            else:
                if cur:
                    pieces.append(cur)
                cur = part
        if cur:
            pieces.append(cur)
        return pieces

    def _infer_low_vram(self, run_piece, text, output_path, interval_silence, verbose):
        """The `low_vram` branch of `infer()` (infer_v2_5.py:510-547; a < 10 GB device there, never an MI355X: `self.low_vram` is False
        unless the caller sets it): the text is cut at punctuation into <= 40-character pieces, each synthesised by its own
        `infer_generator` call with no inner silence, the int16 results joined with `interval_silence` ms of zeros."""
        pieces = self.split_text_by_punctuation(text, max_chars=40)
        if verbose:
            print(f">> Low-VRAM: split into {len(pieces)} segments: {pieces}")
        sr, wavs = 22050, []
        for piece in pieces:
            result = None
            for result in run_piece(piece):
                pass
            if result is not None and isinstance(result, tuple):
                wavs.append(torch.from_numpy(result[1].T).to(torch.int16))
        if not wavs:
            return None
        sil = torch.zeros(1, int(sr * interval_silence / 1000), dtype=torch.int16)
        parts = []
        for i, w in enumerate(wavs):
            parts.append(w)
            if i < len(wavs) - 1:
                parts.append(sil)
        wav = torch.cat(parts, dim=1)
        if output_path:
            if os.path.isfile(output_path):
                os.remove(output_path)
            if os.path.dirname(output_path) != "":
                os.makedirs(os.path.dirname(output_path), exist_ok=True)
            save_pcm_wav(output_path, wav, sr)
            return output_path
        return (sr, wav.numpy().T)

    def infer(self, spk_audio_prompt, text, output_path, lang, emo_audio_prompt=None, emo_alpha=1.0, emo_vector=None,
              use_emo_text=False, emo_text=None, use_random=False, interval_silence=200, verbose=False,
              max_text_tokens_per_segment=120, stream_return=False, more_segment_before=0, duration_factor=1.0,
              text_normalization=True, **generation_kwargs):
        if getattr(self, "low_vram", False) and not stream_return and len(text) > 40:
            return self._infer_low_vram(
                lambda piece: self.infer_generator(spk_audio_prompt, piece, None, lang, emo_audio_prompt, emo_alpha, emo_vector,
                                                   use_emo_text, emo_text, use_random, 0, verbose, max_text_tokens_per_segment, False, 0,
                                                   duration_factor=duration_factor, text_normalization=text_normalization,
                                                   **generation_kwargs),
                text, output_path, interval_silence, verbose)
        gen = self.infer_generator(spk_audio_prompt, text, output_path, lang, emo_audio_prompt, emo_alpha, emo_vector,
                                   use_emo_text, emo_text, use_random, interval_silence, verbose,
                                   max_text_tokens_per_segment, stream_return, more_segment_before, duration_factor,
                                   text_normalization, **generation_kwargs)
        if stream_return:
            return gen
        try:
            return list(gen)[0]
        except IndexError:
            return None

    def infer_generator(self, spk_audio_prompt, text, output_path, lang, emo_audio_prompt=None, emo_alpha=1.0,
                        emo_vector=None, use_emo_text=False, emo_text=None, use_random=False, interval_silence=200,
                        verbose=False, max_text_tokens_per_segment=120, stream_return=False, quick_streaming_tokens=0,
                        duration_factor=1.0, text_normalization=True, **generation_kwargs):
        yield from self._infer_impl(spk_audio_prompt, text, output_path, lang, emo_audio_prompt, emo_alpha, emo_vector, use_emo_text,
                                    emo_text, use_random, interval_silence, verbose, max_text_tokens_per_segment, stream_return,
                                    duration_factor, text_normalization, generation_kwargs)

    def _infer_impl(self, spk_audio_prompt, text, output_path, lang, emo_audio_prompt, emo_alpha, emo_vector, use_emo_text, emo_text,
                    use_random, interval_silence, verbose, max_text_tokens_per_segment, stream_return, duration_factor,
                    text_normalization, generation_kwargs):
        """Body of `infer_generator` (infer_v2_5.py:570-899; infer_v2.py:396-720 has the same flow without `lang`, `duration_factor`
        and `text_normalization`: the v2 subclass passes lang=None, 1.0, True)."""
        start_time = time.perf_counter()
        self._set_gr_progress(0, "starting inference...")
        if use_emo_text:
            raise RuntimeError("use_emo_text=True requires QwenEmotion, but it was not loaded at init "
                               "(use_qwen_emo=False). Re-construct IndexTTS2 with use_qwen_emo=True.")
        if emo_vector is not None:
            emo_audio_prompt = None
            scale = max(0.0, min(1.0, emo_alpha))
            if scale != 1.0:
                emo_vector = [int(x * scale * 10000) / 10000 for x in emo_vector]
        if emo_audio_prompt is None:
            emo_audio_prompt, emo_alpha = spk_audio_prompt, 1.0
        bundle = self._speaker(spk_audio_prompt)
        emovec = self._emovec(bundle, emo_audio_prompt, emo_alpha, emo_vector, use_random)
        self._set_gr_progress(0.1, "text processing...")
        capacity = self.gpt.n_text_pos
        segments = self.frontend.text_segments(text, lang, max_text_tokens_per_segment, text_normalization, capacity)
        if not segments:
            return
        sr = 22050
        lang_id = self.frontend.lang_id(lang) if lang is not None else 0
        wavs = self._synthesize(segments, [lang_id] * len(segments), bundle, emovec, duration_factor,
                                generation_kwargs, max_text_tokens_per_segment)
        silence = None
        if stream_return:
            for w in wavs:
                yield w
                if silence is None:
                    silence = self.interval_silence(wavs, sampling_rate=sr, interval_silence=interval_silence)
                yield silence
        end_time = time.perf_counter()
        self._set_gr_progress(0.9, "saving audio...")
        wav = torch.cat(self.insert_interval_silence(wavs, sampling_rate=sr, interval_silence=interval_silence), dim=1)
        wav_length = wav.shape[-1] / sr
        t = self.last_timing
        print(f">> gpt_gen_time: {t.get('gpt', 0):.2f} seconds")
        print(f">> s2mel_time: {t.get('s2mel', 0):.2f} seconds")
        print(f">> bigvgan_time: {t.get('bigvgan', 0):.2f} seconds")
        print(f">> Total inference time: {end_time - start_time:.2f} seconds")
        print(f">> Generated audio length: {wav_length:.2f} seconds")
        print(f">> RTF: {(end_time - start_time) / max(wav_length, 1e-9):.4f}")
        if output_path:
            if os.path.isfile(output_path):
                os.remove(output_path)
            if os.path.dirname(output_path) != "":
                os.makedirs(os.path.dirname(output_path), exist_ok=True)
            save_pcm_wav(output_path, wav, sr)
            if stream_return:
                return
            yield output_path
        else:
            if stream_return:
                return
            yield (sr, wav.type(torch.int16).numpy().T)

    def infer_batch(self, spk_audio_prompt, texts: Sequence[str], lang, emo_audio_prompt=None, emo_alpha=1.0,
                    interval_silence=200, max_text_tokens_per_segment=120, duration_factor=1.0, text_normalization=True,
                    **generation_kwargs):
        """New capability (BASELINE.json configs[1,2]): many utterances of one speaker in one pass.  Every segment of
        every utterance is a row of one GPT batch / one ragged BigVGAN batch.  Returns a list of (22050, int16 (T,1)).

        Under `torch.distributed` (one process per GPU, `indextts_amd.dist`) the segment rows are LPT-sharded over the ranks
        by text length: rank 0 runs the prompt encoders and broadcasts the speaker bundle (the one collective before the
        decode loop), every rank decodes and vocodes its own rows, the int16 waveforms are gathered on rank 0, which returns
        the full list; the other ranks return None per utterance."""
        from . import dist as D
        world, rank = D.world(), D.rank()
        if emo_audio_prompt is None:
            emo_audio_prompt, emo_alpha = spk_audio_prompt, 1.0
        bundle = None
        if rank == 0:
            bundle = dict(self._speaker(spk_audio_prompt))
            bundle["emo_vec"] = self._emovec(bundle, emo_audio_prompt, emo_alpha, None, False)
        if world > 1:
            bundle = D.broadcast_speaker_bundle(bundle, src=0, device=self.device)
        emovec = bundle["emo_vec"]
        capacity = self.gpt.n_text_pos
        seg_tokens, owner = [], []
        for u, text in enumerate(texts):                    # tokenisation is deterministic host work: every rank does it
            segs = self.frontend.text_segments(text, lang, max_text_tokens_per_segment, text_normalization, capacity)
            seg_tokens += segs
            owner += [u] * len(segs)
        if not seg_tokens:
            return [None] * len(texts)
        mine = D.shard_utterances(len(seg_tokens), rank, world, lengths=[int(t.numel()) for t in seg_tokens]) if world > 1 \
            else list(range(len(seg_tokens)))
        wavs = []
        if mine:
            wavs = self._synthesize([seg_tokens[i] for i in mine], [self.frontend.lang_id(lang)] * len(mine), bundle, emovec,
                                    duration_factor, generation_kwargs, max_text_tokens_per_segment)
        if world > 1:
            wavs = D.gather_waveforms([w.type(torch.int16) for w in wavs], mine, len(seg_tokens), dst=0)
            if rank != 0:
                return [None] * len(texts)
            wavs = [w.float() for w in wavs]                 # int16 -> float is exact; silence is inserted in float
        out = []
        for u in range(len(texts)):
            seg = [w for w, o in zip(wavs, owner) if o == u]
            if not seg:
                out.append(None)
                continue
            wav = torch.cat(self.insert_interval_silence(seg, 22050, interval_silence), dim=1)
            out.append((22050, wav.type(torch.int16).numpy().T))
        return out

    def infer_stream(self, spk_audio_prompt, texts: Sequence[str], lang, emo_audio_prompt=None, emo_alpha=1.0, chunk_size: int = 100,
                     overlap_size: int = 20, max_text_tokens_per_segment=120, duration_factor=1.0, text_normalization=True,
                     **generation_kwargs):
        """Streaming synthesis of a batch of single-segment texts (`FasterIndexTTS2` streaming path,
        backends/trt/pipeline/streaming.py): a generator of `(22050, [int16 array | None per text], [done per text])`, one item per
        GPT chunk of `chunk_size` codes (consecutive chunks share `overlap_size` codes and are cross-faded).  The first audio
        leaves after prefill + `chunk_size` decode steps + one chunk of codes -> mel -> waveform instead of after the whole
        utterance.  Sampling uses one hypothesis per row (`num_beams` is forced to 1: a beam's prefix is not final mid-search)."""
        from .streaming import StreamingDecoder
        if emo_audio_prompt is None:
            emo_audio_prompt, emo_alpha = spk_audio_prompt, 1.0
        bundle = dict(self._speaker(spk_audio_prompt))
        emovec = self._emovec(bundle, emo_audio_prompt, emo_alpha, None, False)
        seg_tokens = []
        for text in texts:
            segs = self.frontend.text_segments(text, lang, max_text_tokens_per_segment, text_normalization, self.gpt.n_text_pos)
            if len(segs) != 1:
                raise ValueError("infer_stream takes texts of one segment each (split long texts with the frontend first)")
            seg_tokens.append(segs[0])
        gk = dict(generation_kwargs)
        gk.pop("do_sample", None)
        gk.pop("num_beams", None)
        max_mel_tokens = gk.pop("max_mel_tokens", 1500)
        gen = dict(do_sample=True, top_p=gk.pop("top_p", 0.8), top_k=gk.pop("top_k", 30), temperature=gk.pop("temperature", 0.8),
                   repetition_penalty=gk.pop("repetition_penalty", 10.0), length_penalty=gk.pop("length_penalty", 0.0), num_beams=1, **gk)
        dev = self.device
        L = max(int(t.numel()) for t in seg_tokens)
        text_ids = torch.full((len(seg_tokens), L), 1, dtype=torch.int32)
        for i, t in enumerate(seg_tokens):
            text_ids[i, : t.numel()] = t.reshape(-1).to(torch.int32)
        langs = torch.tensor([self.frontend.lang_id(lang)] * len(seg_tokens), dtype=torch.long)
        inputs_embeds, attention_mask, max_new, hf = self.gpt.inference_speech_stream(
            bundle["spk_cond_emb"], text_ids.to(dev), chunk_size, overlap_size, langs=langs.to(dev), emo_vec=emovec,
            campplus_embedding=bundle["style"], max_generate_length=max_mel_tokens, **gen)
        up = self.bigvgan.total_up

        def codes_to_audio(codes, code_lens):
            lens = torch.as_tensor(code_lens).to(torch.int32).cpu().clamp(min=1)          # finished rows render one frame pair, dropped by the decoder
            if self.s2mel is not None and self.semantic_codec is not None:
                mel, mel_lens = self.codes_to_mel(codes, lens, bundle, duration_factor)
            else:
                mel, mel_lens = self.frontend.codes_to_mel(codes, lens, bundle, duration_factor)
            wav = self.bigvgan(mel.float(), lens=mel_lens)
            return [wav[i, 0, : int(mel_lens[i]) * up].float().cpu().numpy() for i in range(wav.shape[0])]

        # the cross-fade spans the samples the overlapping codes render to: int(2 * n * 1.72 * duration_factor) frames for n codes here
        # (infer_v2_5.py:833), not the reference decoder's fixed 1.72 frames per code of IndexTTS-2
        dec = StreamingDecoder(self.gpt, codes_to_audio, chunk_size=chunk_size, overlap_size=overlap_size,
                               frames_per_code=2 * 1.72 * float(duration_factor))
        self.last_stream = dec
        yield from dec.generate(inputs_embeds, attention_mask, max_new, **hf)

    def codes_to_mel(self, codes: torch.Tensor, code_lens: torch.Tensor, bundle, duration_factor: float = 1.0,
                     diffusion_steps: int = 25, inference_cfg_rate: float = 0.7, noise: Optional[torch.Tensor] = None):
        """infer_v2_5.py:830-846 for a whole batch of segments on the HIP engine: semantic_codec.decode -> length_regulator ->
        [prompt_condition | cond] -> cfm.inference -> drop the prompt frames.  Every row is processed at its own lengths (what the
        reference's batch-1 call per segment computes).  Returns mel (B, 80, max frames) f32 and the frame counts (B,) int32."""
        from .s2mel import codes_to_mel
        return codes_to_mel(self.semantic_codec, self.s2mel.models, codes, code_lens, bundle, duration_factor, diffusion_steps,
                            inference_cfg_rate, noise)

    # ---- the hot path: one GPT batch, one ragged vocoder batch -------------------------------------------------------
    def _synthesize(self, segment_tokens: List[torch.Tensor], lang_ids: List[int], bundle, emovec, duration_factor,
                    generation_kwargs, max_text_tokens_per_segment) -> List[torch.Tensor]:
        gk = dict(generation_kwargs)
        gk.pop("do_sample", None)                       # popped and ignored by the reference too (:732,781)
        top_p, top_k = gk.pop("top_p", 0.8), gk.pop("top_k", 30)
        temperature = gk.pop("temperature", 0.8)
        length_penalty = gk.pop("length_penalty", 0.0)
        num_beams = gk.pop("num_beams", 3)
        repetition_penalty = gk.pop("repetition_penalty", 10.0)
        max_mel_tokens = gk.pop("max_mel_tokens", 1500)
        dev = self.device
        L = max(int(t.numel()) for t in segment_tokens)
        text = torch.full((len(segment_tokens), L), 1, dtype=torch.int32)      # stop_text_token right padding (:726)
        for i, t in enumerate(segment_tokens):
            text[i, : t.numel()] = t.reshape(-1).to(torch.int32)
        langs = torch.tensor(lang_ids, dtype=torch.long)
        inflight_slots = gk.pop("inflight_slots", None)    # engine extension: decode `inflight_slots` rows at a time, admit waiting segments into
        inflight_kw = {k: gk.pop(k) for k in ("chunk_tokens", "min_free") if k in gk}              # slots whose row has stopped
        t0 = time.perf_counter()
        if inflight_slots and num_beams == 1 and len(segment_tokens) > int(inflight_slots):
            codes, _ = self.gpt.inference_speech_inflight(bundle["spk_cond_emb"], text.to(dev), langs.to(dev), emo_vec=emovec,
                                                          campplus_embedding=bundle["style"], do_sample=True, top_p=top_p, top_k=top_k,
                                                          temperature=temperature, length_penalty=length_penalty, num_beams=1,
                                                          repetition_penalty=repetition_penalty, max_generate_length=max_mel_tokens,
                                                          slots=int(inflight_slots), **inflight_kw, **gk)
        else:
            codes, _ = self.gpt.inference_speech(bundle["spk_cond_emb"], text.to(dev), langs.to(dev), emo_vec=emovec,
                                                 campplus_embedding=bundle["style"], do_sample=True, top_p=top_p, top_k=top_k,
                                                 temperature=temperature, num_return_sequences=1, length_penalty=length_penalty,
                                                 num_beams=num_beams, repetition_penalty=repetition_penalty,
                                                 max_generate_length=max_mel_tokens, **gk)
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        t1 = time.perf_counter()
        if codes.shape[1] > 0 and (codes[:, -1] != self.stop_mel_token).any():
            warnings.warn(f"WARN: generation stopped due to exceeding `max_mel_tokens` ({max_mel_tokens}). "
                          f"Consider reducing `max_text_tokens_per_segment`({max_text_tokens_per_segment}) or increasing "
                          f"`max_mel_tokens`.", category=RuntimeWarning)
        codes, code_lens = self.trim_codes(codes)
        if self.s2mel is not None and self.semantic_codec is not None:
            mel, mel_lens = self.codes_to_mel(codes, code_lens, bundle, duration_factor)
        else:
            mel, mel_lens = self.frontend.codes_to_mel(codes, code_lens, bundle, duration_factor)
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        t2 = time.perf_counter()
        wav = self.bigvgan(mel.float(), lens=mel_lens)                      # (B, 1, Tmax*256), rows bounded at own length
        wav = torch.clamp(PCM16_MAX * wav, -PCM16_MAX, PCM16_MAX)           # (:855)
        up = self.bigvgan.total_up
        out = [wav[i, :, : int(mel_lens[i]) * up].cpu() for i in range(wav.shape[0])]
        t3 = time.perf_counter()
        self.last_timing = dict(gpt=t1 - t0, s2mel=t2 - t1, bigvgan=t3 - t2)
        return out


W2V_TAP_LAYER = 17                    # `hidden_states[17]` of the w2v-bert-2.0 encoder (infer_v2_5.py:288)


class ReferenceFrontend(Frontend):
    """Prompt / text stages on the reference's own PyTorch modules.

    The reference package builds every prompt-side module in one place, `indextts.infer_v2_5.IndexTTS2.__init__`
    (:117-266: w2v-bert feature extractor + model + statistics, semantic codec, s2mel, CAMPPlus, tokenizer, text normaliser,
    emotion / speaker matrices, mel function).  Instead of restating that loader, this frontend constructs the reference
    object itself from the same `cfg_path` / `model_dir` (`ref=` injects an existing one) and drives its modules with the call
    sequences of `infer_generator` (:620-727).  The reference's GPT transformer stack and vocoder are not used (the engine
    replaces them) and are dropped after construction; its emotion encoder (`gpt.merge_emovec`) is kept.

    Needs the reference `indextts` package with its third-party dependencies (torchaudio, librosa, transformers, ...) and the
    checkpoint directory: where those are missing the constructor raises ImportError / FileNotFoundError -- there is no fallback.
    """

    def __init__(self, cfg, model_dir, device, gpt_engine=None, cfg_path=None, ref=None):
        if ref is None:
            try:
                from indextts.infer_v2_5 import IndexTTS2 as RefIndexTTS2
            except Exception as e:                                   # loud: there is no fallback
                raise ImportError("ReferenceFrontend needs the reference `indextts` package and its prompt-side dependencies "
                                  f"(torchaudio, librosa, transformers, ...): {e!r}.  Inject frontend= instead, e.g. "
                                  "indextts_amd.frontend.EngineFrontend(cfg, model_dir, device, text_frontend=...), which reads the "
                                  "checkpoint directory itself and needs no reference package.") from e
            ref = RefIndexTTS2(cfg_path=cfg_path or os.path.join(model_dir, "config.yaml"), model_dir=model_dir, use_bf16=False,
                               device=device, use_cuda_kernel=False)
            for name in ("bigvgan",):                                # replaced by the engine
                if hasattr(ref, name):
                    setattr(ref, name, None)
            if getattr(ref, "gpt", None) is not None:
                for name in ("gpt", "inference_model"):              # the GPT-2 stack; merge_emovec / get_emovec stay
                    if hasattr(ref.gpt, name):
                        setattr(ref.gpt, name, None)
        self.ref = ref
        self.cfg, self.model_dir, self.device = cfg, model_dir, device
        self.tokenizer = ref.tokenizer
        # the CAMPPlus speaker encoder runs on the engine (indextts_amd/campplus.py) from the weights the reference module loaded;
        # the reference's own module stays in place only when its configuration is not the shipped one
        self.campplus = None
        cm = getattr(ref, "campplus_model", None)
        if cm is not None and hasattr(cm, "state_dict") and torch.cuda.is_available() and str(device).startswith("cuda"):
            try:
                from .campplus import CAMPPlus
                self.campplus = CAMPPlus(feat_dim=80, embedding_size=192, device=device).load_state_dict(cm.state_dict())
            except NotImplementedError:
                self.campplus = None

        # likewise the w2v-bert-2.0 feature encoder (indextts_amd/w2vbert.py): only the 17 layers below the tapped hidden state
        self.w2v = None
        sm = getattr(ref, "semantic_model", None)
        if sm is not None and hasattr(sm, "config") and torch.cuda.is_available() and str(device).startswith("cuda"):
            try:
                from .w2vbert import Wav2Vec2BertModel
                c = sm.config
                self.w2v = Wav2Vec2BertModel(
                    hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                    intermediate_size=c.intermediate_size, feature_projection_input_dim=c.feature_projection_input_dim,
                    position_embeddings_type=c.position_embeddings_type, left_max_position_embeddings=c.left_max_position_embeddings,
                    right_max_position_embeddings=c.right_max_position_embeddings, conv_depthwise_kernel_size=c.conv_depthwise_kernel_size,
                    hidden_act=c.hidden_act, layer_norm_eps=c.layer_norm_eps, add_adapter=c.add_adapter, device=device,
                ).load_state_dict(sm.state_dict(), n_layers=W2V_TAP_LAYER)
            except NotImplementedError:
                self.w2v = None

        # and the DSP in front of them (indextts_amd/audio.py: resampling, SeamlessM4T features, prompt log-mel, Kaldi fbank): the engine
        # versions take over when the s2mel spectrogram parameters are the ones the kernel supports (win_length == n_fft, center=False)
        self.audio = None
        if torch.cuda.is_available() and str(device).startswith("cuda"):
            try:
                from . import audio as A
                sp = cfg["s2mel"]["preprocess_params"]
                spect = sp["spect_params"]
                fmax = spect.get("fmax", "None") if hasattr(spect, "get") else "None"
                self._mel_args = dict(n_fft=int(spect["n_fft"]), win_size=int(spect["win_length"]), hop_size=int(spect["hop_length"]),
                                      num_mels=int(spect["n_mels"]), sampling_rate=int(sp["sr"]),
                                      fmin=spect.get("fmin", 0) if hasattr(spect, "get") else 0,
                                      fmax=None if fmax == "None" else 8000, center=False)          # infer_v2_5.py:256-265
                if self._mel_args["win_size"] == self._mel_args["n_fft"]:
                    self.audio = A
                    self._features = A.SeamlessM4TFeatureExtractor(device=device)
            except (KeyError, TypeError, AttributeError):
                self.audio = None

    # ---- the reference's own state dicts, for the engine stages (IndexTTS2.__init__ loads them when none are injected) --------
    def engine_state_dicts(self):
        return dict(semantic_codec=self.ref.semantic_codec.state_dict(), cfm=self.ref.s2mel.models["cfm"].state_dict(),
                    length_regulator=self.ref.s2mel.models["length_regulator"].state_dict())

    # ---- infer_generator :620-667 -------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _w2v(self, audio_16k):
        r = self.ref
        inputs = (self._features if self.audio is not None else r.extract_features)(audio_16k, sampling_rate=16000, return_tensors="pt")
        if self.w2v is not None:
            return self.w2v.get_emb(inputs["input_features"], inputs["attention_mask"], r.semantic_mean, r.semantic_std, layer=W2V_TAP_LAYER)
        return r.get_emb(inputs["input_features"].to(self.device), inputs["attention_mask"].to(self.device))

    @torch.no_grad()
    def speaker_bundle(self, spk_audio_prompt):
        r = self.ref
        audio, sr = r._load_and_cut_audio(spk_audio_prompt, 15, False)
        if self.audio is not None:                                   # DSP on the engine (indextts_amd/audio.py)
            A = self.audio
            audio_22k = A.Resample(sr, 22050, device=self.device)(audio)
            audio_16k = A.Resample(sr, 16000, device=self.device)(audio)
            spk_cond_emb = self._w2v(audio_16k)
            ref_mel = A.mel_spectrogram(audio_22k, **self._mel_args)
            feat = A.subtract_mean(A.fbank(audio_16k, num_mel_bins=80, dither=0, sample_frequency=16000))
        else:
            import torchaudio
            audio_22k = torchaudio.transforms.Resample(sr, 22050)(audio)
            audio_16k = torchaudio.transforms.Resample(sr, 16000)(audio)
            spk_cond_emb = self._w2v(audio_16k)
            ref_mel = r.mel_fn(audio_22k.to(spk_cond_emb.device).float())
            feat = torchaudio.compliance.kaldi.fbank(audio_16k.to(ref_mel.device), num_mel_bins=80, dither=0, sample_frequency=16000)
            feat = feat - feat.mean(dim=0, keepdim=True)
        style = (self.campplus or r.campplus_model)(feat.unsqueeze(0))
        prompt_condition = r.s2mel.models["length_regulator"](spk_cond_emb, ylens=torch.LongTensor([ref_mel.size(2)]).to(ref_mel.device),
                                                              n_quantizers=3, f0=None)[0]
        return dict(style=style, spk_cond_emb=spk_cond_emb, ref_mel=ref_mel, prompt_condition=prompt_condition)

    @torch.no_grad()
    def emo_cond(self, emo_audio_prompt):                            # :682-697
        emo_audio, _ = self.ref._load_and_cut_audio(emo_audio_prompt, 15, False, sr=16000)
        return self._w2v(emo_audio)

    @torch.no_grad()
    def merge_emovec(self, spk_cond_emb, emo_cond_emb, alpha):       # :759-765 (the lengths are the feature dims there)
        dev = spk_cond_emb.device
        return self.ref.gpt.merge_emovec(spk_cond_emb, emo_cond_emb, torch.tensor([spk_cond_emb.shape[-1]], device=dev),
                                         torch.tensor([emo_cond_emb.shape[-1]], device=dev), alpha=alpha)

    def emo_vector_mix(self, emo_vector, style, use_random):         # :669-680
        import random
        r = self.ref
        from indextts.infer_v2_5 import find_most_similar_cosine
        w = torch.tensor(emo_vector, device=self.device)
        idx = [random.randint(0, x - 1) for x in r.emo_num] if use_random else [find_most_similar_cosine(style, t) for t in r.spk_matrix]
        mat = torch.cat([t[i].unsqueeze(0) for i, t in zip(idx, r.emo_matrix)], 0)
        return torch.sum(w.unsqueeze(1) * mat, 0).unsqueeze(0), torch.sum(w)

    def text_segments(self, text, lang, max_text_tokens_per_segment, text_normalization, capacity):     # :699-726
        import re
        from indextts import infer_v2_5 as R
        r, lo = self.ref, lang.lower()
        prefix = f"<|{lo}|> "
        text = r.text_process.clean_pattern.sub(lambda x: r.text_process.char_rep_map[x.group()], text)
        if text_normalization:
            if lo in ("zh", "zhen", "en"):
                text = r.text_process.normalize(text)
            elif lo in ("ja", "es"):
                text = R.nemo_text_normalize(text, lo)
        if lo in ("ja", "zh", "zhen", "en"):
            text = text.lower()
        if lo == "es":
            text = text.upper()
        text = R.apply_pronunciation_annotations(text)
        if lo == "ja":
            text = r.ja_text_process.process_ja_text(text)
        text = re.sub(r"<\|([^|]+)\|>", lambda m: f"<|{m.group(1).upper()}|>", text)
        out = []
        for seg in r.split_text_by_tokens(text, max_text_tokens_per_segment, prefix):
            toks = r.tokenizer.encode(prefix + seg, allowed_special="all")
            out.append(torch.tensor(list(toks) + [1], dtype=torch.int32))          # F.pad(toks, (0, 1), value=1)
        return out

    def lang_id(self, lang):
        from indextts.utils.tokenizer import lang_to_token
        return int(lang_to_token(lang))

    @torch.no_grad()
    def codes_to_mel(self, codes, code_lens, bundle, duration_factor):            # :830-846, segment by segment at batch 1
        r = self.ref
        mels, lens = [], []
        for b in range(codes.shape[0]):
            n = int(code_lens[b])
            S_infer = r.semantic_codec.decode(codes[b:b + 1, :n])
            target = torch.LongTensor([int(S_infer.shape[1] * 1.72 * duration_factor)]).to(codes.device)
            cond = r.s2mel.models["length_regulator"](S_infer, ylens=target, n_quantizers=3, f0=None)[0]
            cat = torch.cat([bundle["prompt_condition"], cond], dim=1)
            mel = r.s2mel.models["cfm"].inference(cat, torch.LongTensor([cat.size(1)]).to(cond.device), bundle["ref_mel"], bundle["style"],
                                                  None, 25, inference_cfg_rate=0.7)[:, :, bundle["ref_mel"].size(-1):]
            mels.append(mel.float())
            lens.append(mel.shape[-1])
        out = torch.zeros(len(mels), mels[0].shape[1], max(lens), device=codes.device)
        for b, m in enumerate(mels):
            out[b, :, : lens[b]] = m[0]
        return out, torch.tensor(lens, dtype=torch.int32)
