"""`IndexTTS` (v1 / v1.5) pipeline class with the reference's constructor / `infer()` / `infer_fast()` signatures, the
GPT decode, the teacher-forced latent pass and the speaker-conditioned BigVGAN re-routed to the HIP engine.

Reference mirrored: `indextts/infer.py::IndexTTS` (`__init__` :29-133, `remove_long_silence` :135-190, `bucket_segments`
:192-248, `pad_tokens_cat` :250-268, `infer_fast` :284-517, `infer` :520-688).  Differences behind the API:
  * `self.gpt` is `indextts_amd.gpt.UnifiedVoiceV1`, `self.bigvgan` is `indextts_amd.bigvgan.BigVGAN` (v1 variant:
    speaker-conditioning biases, tanh);
  * `infer()` keeps the reference's segment-by-segment order of operations; `infer_fast()` keeps its bucketing, the
    B=1 latent passes and the 2-latents-per-vocoder-call chunking;
  * audio loading + mel features, the text normaliser/tokenizer and the Conformer+Perceiver conditioning encoder are NOT on
    the hot path and come from a `frontend` object (`FrontendV1`); with `frontend=None` the reference package is required
    and its absence is an error, never a fallback.
"""
import os
import time
import warnings
from typing import Dict, List, Optional

import torch
from torch.nn.utils.rnn import pad_sequence

from .infer_v2_5 import save_pcm_wav


class FrontendV1:
    """What the v1 pipeline consumes from outside the hot path."""

    tokenizer = None        # .tokenize(text) -> tokens, .split_segments(tokens, max_text_tokens_per_segment), .convert_tokens_to_ids

    def cond_mel(self, audio_prompt, truncate_seconds: Optional[float] = None) -> torch.Tensor:
        """load, mono, resample to 24 kHz, `MelSpectrogramFeatures` -> (1, 100, T)  (infer.py:303-323,529-537)"""
        raise NotImplementedError

    def conditioning(self, cond_mel: torch.Tensor, cond_mel_lengths: torch.Tensor) -> torch.Tensor:
        """`UnifiedVoice.get_conditioning` (Conformer + Perceiver) -> (1, 32, D)  (gpt/model.py:495-524)"""
        raise NotImplementedError


class IndexTTS:
    def __init__(self, cfg_path="checkpoints/config.yaml", model_dir="checkpoints", use_fp16=True, device=None,
                 use_cuda_kernel=None, *, frontend: Optional[FrontendV1] = None, gpt=None, bigvgan=None,
                 cfg: Optional[dict] = None):
        if device is not None:
            self.device = device
        elif torch.cuda.is_available():
            self.device = "cuda:0"
        else:
            raise RuntimeError("IndexTTS (HIP engine) needs a GPU: there is no CPU path")
        # fp16 autocast of the reference maps to the engine's bf16 weight/KV mode; the vocoder stays f32
        self.use_fp16 = bool(use_fp16)
        self.use_cuda_kernel = True
        self.dtype = torch.bfloat16 if self.use_fp16 else None
        self.model_dir = model_dir
        if cfg is None:
            import yaml
            with open(cfg_path) as f:
                cfg = yaml.safe_load(f)
        self.cfg = cfg
        gcfg = dict(cfg["gpt"])
        self.stop_mel_token = gcfg.get("stop_mel_token", 8193)
        self.model_version = cfg.get("version", None)
        if frontend is None:
            raise RuntimeError("IndexTTS: a frontend (audio features, tokenizer, conditioning encoder) is required; the "
                               "reference package that provides them is not importable in this environment")
        self.frontend = frontend
        self.tokenizer = frontend.tokenizer
        if gpt is None:
            from .gpt import UnifiedVoiceV1
            gpt = UnifiedVoiceV1(**gcfg, precision="bf16" if self.use_fp16 else "fp32", device=self.device)
            ck = torch.load(os.path.join(model_dir, cfg["gpt_checkpoint"]), map_location="cpu")
            gpt.load_state_dict(ck["model"] if "model" in ck else ck)
            gpt.post_init_gpt2_config(use_deepspeed=False, kv_cache=True, half=self.use_fp16)
        if getattr(gpt, "conditioning_fn", None) is None:
            gpt.conditioning_fn = frontend.conditioning
        self.gpt = gpt
        if bigvgan is None:
            raise RuntimeError("IndexTTS: pass bigvgan= (indextts_amd.bigvgan.BigVGAN built with cond_dim/in_channels and "
                               "the reference ECAPA speaker encoder, see INTEGRATION.md)")
        self.bigvgan = bigvgan
        self.cache_audio_prompt = None
        self.cache_cond_mel = None
        self.gr_progress = None
        self.last_timing: Dict[str, float] = {}

    # ---- helpers with the reference's semantics ------------------------------------------------------------------
    def _set_gr_progress(self, value, desc):
        if self.gr_progress is not None:
            self.gr_progress(value, desc=desc)

    def torch_empty_cache(self):
        pass                                    # the engine's workspaces are reused, nothing to trim

    def remove_long_silence(self, codes: torch.Tensor, silent_token=52, max_consecutive=30):
        """infer.py:135-190.  Per row: length = index of the first stop token (or the row length); when a row holds
        more than `max_consecutive` silent tokens IN TOTAL, every run of silent tokens is cut to 10; rows are re-padded
        with the stop token and clipped to the longest kept length.  Returns (codes, code_lens)."""
        device = codes.device
        rows = codes.detach().to("cpu")
        kept: List[torch.Tensor] = []
        lens: List[int] = []
        changed = False
        for row in rows:
            stop = (row == self.stop_mel_token).nonzero()
            n = int(stop[0]) if stop.numel() else row.numel()
            if int((row == silent_token).sum()) > max_consecutive:
                body = row[:n]
                sil = body == silent_token
                # position inside the current silent run (1-based), 0 for non-silent tokens
                idx = torch.arange(n)
                run_start = torch.where(~sil, idx, torch.full_like(idx, -1)).cummax(0).values
                run_pos = idx - run_start
                keep = ~sil | (run_pos <= 10)
                row_k = body[keep]
                kept.append(row_k)
                lens.append(int(row_k.numel()))
                changed = True
            else:
                kept.append(row[:n])
                lens.append(n)
        if changed:
            out = pad_sequence(kept, batch_first=True, padding_value=self.stop_mel_token) if len(kept) > 1 else kept[0][None]
        else:
            out = rows
        m = max(lens)
        if m < out.shape[1]:
            out = out[:, :m]
        return out.to(device), torch.tensor(lens, dtype=torch.long, device=device)

    def bucket_segments(self, segments, bucket_max_size=4) -> List[List[Dict]]:
        """infer.py:192-248: length-sorted buckets (new bucket when the length reaches 1.5x the bucket median or the
        bucket is full), then singleton buckets are merged into buckets with room / grouped together."""
        items = [{"idx": i, "sent": s, "len": len(s)} for i, s in enumerate(segments)]
        if len(items) <= bucket_max_size:
            return [items]
        buckets: List[List[Dict]] = []
        median = 0
        for it in sorted(items, key=lambda x: x["len"]):
            if it["len"] == 0:
                continue
            cur = buckets[-1] if buckets else None
            if cur is None or it["len"] >= int(median * 1.5) or len(cur) >= bucket_max_size:
                buckets.append([it])
                median = it["len"]
            else:
                cur.append(it)
                median = cur[len(cur) // 2]["len"]
        multi = [b for b in buckets if len(b) > 1]
        ones = [b[0] for b in buckets if len(b) == 1]
        for b in multi:
            if not ones:
                break
            if len(b) < bucket_max_size:
                b.append(ones.pop(0))
        multi.extend(ones[i:i + bucket_max_size] for i in range(0, len(ones), bucket_max_size))
        return multi

    def pad_tokens_cat(self, tokens: List[torch.Tensor]) -> torch.Tensor:
        """infer.py:250-268: v1.5+: right-pad with stop_text; v1.0: up to 8 stop_text then start_text ids."""
        stop_t = self.cfg["gpt"].get("stop_text_token", 1)
        start_t = self.cfg["gpt"].get("start_text_token", 0)
        if self.model_version and float(self.model_version) >= 1.5:
            return pad_sequence([t.squeeze(0) for t in tokens], batch_first=True, padding_value=stop_t)
        max_len = max(t.size(1) for t in tokens)
        out = []
        for t in tokens:
            pad = max_len - t.size(1)
            if pad > 0:
                n = min(8, pad)
                t = torch.nn.functional.pad(t, (0, n), value=stop_t)
                t = torch.nn.functional.pad(t, (0, pad - n), value=start_t)
            out.append(t[:, :max_len])
        return torch.cat(out, dim=0)

    # ---- shared pieces ----------------------------------------------------------------------------------------
    def _cond_mel(self, audio_prompt, truncate_seconds=None):
        if self.cache_cond_mel is None or self.cache_audio_prompt != audio_prompt:
            cond_mel = self.frontend.cond_mel(audio_prompt, truncate_seconds).to(self.device)
            self.cache_audio_prompt = audio_prompt
            self.cache_cond_mel = cond_mel
        return self.cache_cond_mel

    @staticmethod
    def _gen_kwargs(generation_kwargs):
        g = dict(generation_kwargs)
        out = dict(do_sample=g.pop("do_sample", True), top_p=g.pop("top_p", 0.8), top_k=g.pop("top_k", 30),
                   temperature=g.pop("temperature", 1.0), length_penalty=g.pop("length_penalty", 0.0),
                   num_beams=g.pop("num_beams", 3), repetition_penalty=g.pop("repetition_penalty", 10.0))
        max_mel_tokens = g.pop("max_mel_tokens", 600)
        out.update(g)
        return out, max_mel_tokens

    def _finish(self, wavs, sampling_rate, output_path):
        wav = torch.cat(wavs, dim=1).cpu()
        if output_path:
            if os.path.isfile(output_path):
                os.remove(output_path)
            if os.path.dirname(output_path) != "":
                os.makedirs(os.path.dirname(output_path), exist_ok=True)
            save_pcm_wav(output_path, wav, sampling_rate)
            return output_path
        return (sampling_rate, wav.type(torch.int16).numpy().T)

    def _warn_overflow(self, max_mel_tokens, max_text_tokens_per_segment):
        warnings.warn(f"WARN: generation stopped due to exceeding `max_mel_tokens` ({max_mel_tokens}). "
                      f"Consider reducing `max_text_tokens_per_segment`({max_text_tokens_per_segment}) or increasing "
                      f"`max_mel_tokens`.", category=RuntimeWarning)

    # ---- infer (infer.py:520-688): one segment at a time -----------------------------------------------------------
    def infer(self, audio_prompt, text, output_path, verbose=False, max_text_tokens_per_segment=120, **generation_kwargs):
        self._set_gr_progress(0, "starting inference...")
        t_start = time.perf_counter()
        cond_mel = self._cond_mel(audio_prompt)
        cond_len = torch.tensor([cond_mel.shape[-1]], device=self.device)
        self._set_gr_progress(0.1, "text processing...")
        tokens = self.tokenizer.tokenize(text)
        segments = self.tokenizer.split_segments(tokens, max_text_tokens_per_segment)
        gk, max_mel_tokens = self._gen_kwargs(generation_kwargs)
        sampling_rate = 24000
        conds = self.gpt.get_conditioning(cond_mel, cond_len)
        wavs, warned = [], False
        t_gen = t_fwd = t_voc = 0.0
        for i, sent in enumerate(segments):
            ids = torch.tensor(self.tokenizer.convert_tokens_to_ids(sent), dtype=torch.int32, device=self.device)[None]
            self._set_gr_progress(0.2 + 0.4 * i / len(segments), f"gpt latents inference {i + 1}/{len(segments)}...")
            t0 = time.perf_counter()
            codes = self.gpt.inference_speech(cond_mel, ids, cond_mel_lengths=cond_len, conds_latent=conds,
                                              max_generate_length=max_mel_tokens, **gk)
            t_gen += time.perf_counter() - t0
            if not warned and bool((codes[:, -1] != self.stop_mel_token).any()):
                self._warn_overflow(max_mel_tokens, max_text_tokens_per_segment)
                warned = True
            codes, code_lens = self.remove_long_silence(codes, silent_token=52, max_consecutive=30)
            t0 = time.perf_counter()
            latent = self.gpt(cond_mel, ids, torch.tensor([ids.shape[-1]], device=self.device), codes,
                              code_lens * self.gpt.mel_length_compression, cond_mel_lengths=cond_len, return_latent=True,
                              clip_inputs=False, conds_latent=conds)
            t_fwd += time.perf_counter() - t0
            t0 = time.perf_counter()
            wav, _ = self.bigvgan(latent, cond_mel.transpose(1, 2))
            t_voc += time.perf_counter() - t0
            wavs.append(torch.clamp(32767 * wav.squeeze(1), -32767.0, 32767.0).cpu())
        self.last_timing = dict(gpt_gen=t_gen, gpt_forward=t_fwd, bigvgan=t_voc, total=time.perf_counter() - t_start)
        self._set_gr_progress(0.9, "saving audio...")
        return self._finish(wavs, sampling_rate, output_path)

    # ---- infer_fast (infer.py:284-517): bucketed batches -------------------------------------------------------------
    def infer_fast(self, audio_prompt, text, output_path, verbose=False, max_text_tokens_per_segment=100,
                   segments_bucket_max_size=4, **generation_kwargs):
        self._set_gr_progress(0, "starting fast inference...")
        t_start = time.perf_counter()
        cond_mel = self._cond_mel(audio_prompt, truncate_seconds=50)
        cond_len = torch.tensor([cond_mel.shape[-1]], device=self.device)
        tokens = self.tokenizer.tokenize(text)
        segments = self.tokenizer.split_segments(tokens, max_text_tokens_per_segment=max_text_tokens_per_segment)
        gk, max_mel_tokens = self._gen_kwargs(generation_kwargs)
        sampling_rate = 24000
        conds = self.gpt.get_conditioning(cond_mel, cond_len)
        self._set_gr_progress(0.1, "text processing...")
        buckets = self.bucket_segments(segments, bucket_max_size=segments_bucket_max_size)
        bucket_tokens = [[torch.tensor(self.tokenizer.convert_tokens_to_ids(it["sent"]), dtype=torch.int32,
                                       device=self.device)[None] for it in b] for b in buckets]
        total = sum(len(b) for b in buckets)
        t_gen = t_fwd = t_voc = 0.0
        bucket_codes, done = [], 0
        for toks in bucket_tokens:
            batch = self.pad_tokens_cat(toks) if len(toks) > 1 else toks[0]
            done += len(toks)
            self._set_gr_progress(0.2 + 0.3 * done / total, f"gpt speech inference {done}/{total}...")
            t0 = time.perf_counter()
            bucket_codes.append(self.gpt.inference_speech(cond_mel, batch, cond_mel_lengths=cond_len, conds_latent=conds,
                                                          max_generate_length=max_mel_tokens, **gk))
            t_gen += time.perf_counter() - t0
        self._set_gr_progress(0.5, "gpt latents inference...")
        latents: Dict[int, torch.Tensor] = {}
        warned = False
        for codes_b, toks, b in zip(bucket_codes, bucket_tokens, buckets):
            for i, it in enumerate(b):
                row = codes_b[i]
                if not warned and int(row[-1]) != self.stop_mel_token:
                    self._warn_overflow(max_mel_tokens, max_text_tokens_per_segment)
                    warned = True
                codes, code_lens = self.remove_long_silence(row[None], silent_token=52, max_consecutive=30)
                # B=1 like the reference (infer.py:446-456): this pass has no attention mask, so right-padding a shorter
                # text row inside a batch would change its latents
                t0 = time.perf_counter()
                ids = toks[i]
                latents[it["idx"]] = self.gpt(cond_mel, ids, torch.tensor([ids.shape[-1]], device=self.device), codes,
                                              code_lens * self.gpt.mel_length_compression, cond_mel_lengths=cond_len,
                                              return_latent=True, clip_inputs=False, conds_latent=conds)
                t_fwd += time.perf_counter() - t0
        ordered = [latents[i] for i in sorted(latents)]
        chunk_size = 2
        self._set_gr_progress(0.7, "bigvgan decoding...")
        wavs = []
        for i in range(0, len(ordered), chunk_size):
            latent = torch.cat(ordered[i:i + chunk_size], dim=1)
            t0 = time.perf_counter()
            wav, _ = self.bigvgan(latent, cond_mel.transpose(1, 2))
            t_voc += time.perf_counter() - t0
            wavs.append(torch.clamp(32767 * wav.squeeze(1), -32767.0, 32767.0).cpu())
        self.last_timing = dict(gpt_gen=t_gen, gpt_forward=t_fwd, bigvgan=t_voc, total=time.perf_counter() - t_start)
        self._set_gr_progress(0.9, "saving audio...")
        return self._finish(wavs, sampling_rate, output_path)
