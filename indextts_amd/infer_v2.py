"""`IndexTTS2` of IndexTTS-2 (`indextts/infer_v2.py`, BASELINE.json configs[3]) with the reference's constructor / `infer()`
signatures and the hot stages on the HIP engine.

Differences from the v2.5 pipeline (`indextts_amd/infer_v2_5.py`, whose batching, caching and output handling are inherited):
  * `self.gpt` is `UnifiedVoice` in the reference's default conditioning mode (34 conditioning tokens: 32 Conformer + Perceiver
    speaker latents + emo_vec, two speed embeddings; infer_v2.py:98, model_v2.py:767-773); the speaker / emotion encoders are
    prompt-side PyTorch modules reached through `gpt.conditioning_fn` and the frontend;
  * after decoding, the teacher-forced latent pass `self.gpt(...)` (infer_v2.py:636-651) runs on the engine
    (`itts_gpt_forward_latent`) and its output goes through `s2mel.models['gpt_layer']`;
  * the content features are `semantic_codec.quantizer.vq2emb(codes) + latent` (no Vocos decode), the mel length is
    `(code_lens * 1.72).long()` (:659-662);
  * the speaker prompt's content condition goes through the codec's `quantize` half first (`prompt_condition`, :465-479).
"""
import time
import warnings
from typing import List

import torch

from .infer_v2_5 import PCM16_MAX, Frontend, IndexTTS2 as _IndexTTS2V25  # noqa: F401


class IndexTTS2(_IndexTTS2V25):
    USE_GPT_LATENT = True
    SPK_COND_MODE = "conformer"            # infer_v2.py:98: `UnifiedVoice(**cfg.gpt)`, the default conditioning mode (no spk_emb_proj.*)

    @staticmethod
    def _bigvgan_dir(model_dir, aux_paths=None):
        """infer_v2.py:176-177: `aux_paths["bigvgan"]`; `ensure_models_available(model_dir)` (utils/model_download.py:213-214) places it
        at `<model_dir>/hf_cache/bigvgan`, which is where it is looked for when no `aux_paths` is given (no downloads here)."""
        import os
        if aux_paths and "bigvgan" in aux_paths:
            return aux_paths["bigvgan"]
        return os.path.join(model_dir, "hf_cache", "bigvgan")

    def __init__(self, cfg_path="checkpoints/config.yaml", model_dir="checkpoints", use_fp16=False, device=None, use_cuda_kernel=None,
                 use_deepspeed=False, use_accel=False, use_torch_compile=False, use_qwen_emo=True, aux_paths=None, *, frontend=None,
                 gpt=None, bigvgan=None, cfg=None, semantic_codec=None, s2mel=None, codes_to_mel="auto"):
        """infer_v2.py:37-41.  `use_fp16` selects the engine's reduced-precision (bf16) GPT mode; QwenEmotion (text -> emotion
        vector) is a prompt-side LLM: when no frontend providing it is injected, `use_emo_text` raises like the reference does
        without the model."""
        super().__init__(cfg_path=cfg_path, model_dir=model_dir, use_bf16=use_fp16, device=device, use_cuda_kernel=use_cuda_kernel,
                         use_deepspeed=use_deepspeed, use_accel=use_accel, use_torch_compile=use_torch_compile, use_qwen_emo=False,
                         frontend=frontend, gpt=gpt, bigvgan=bigvgan, cfg=cfg, semantic_codec=semantic_codec, s2mel=s2mel,
                         aux_paths=aux_paths, codes_to_mel=codes_to_mel)
        self.use_fp16 = bool(use_fp16)
        self.aux_paths = aux_paths
        self.model_version = (cfg or self.cfg).get("version", 2.0)

    # ---- API: the reference v2 signatures (infer_v2.py:371-400) -- no `lang`, `duration_factor`, `text_normalization` -------------
    def infer(self, spk_audio_prompt, text, output_path, emo_audio_prompt=None, emo_alpha=1.0, emo_vector=None, use_emo_text=False,
              emo_text=None, use_random=False, interval_silence=200, verbose=False, max_text_tokens_per_segment=120,
              stream_return=False, more_segment_before=0, **generation_kwargs):
        gen = self.infer_generator(spk_audio_prompt, text, output_path, emo_audio_prompt, emo_alpha, emo_vector, use_emo_text, emo_text,
                                   use_random, interval_silence, verbose, max_text_tokens_per_segment, stream_return,
                                   more_segment_before, **generation_kwargs)
        if stream_return:
            return gen
        try:
            return list(gen)[0]
        except IndexError:
            return None

    def infer_generator(self, spk_audio_prompt, text, output_path, emo_audio_prompt=None, emo_alpha=1.0, emo_vector=None,
                        use_emo_text=False, emo_text=None, use_random=False, interval_silence=200, verbose=False,
                        max_text_tokens_per_segment=120, stream_return=False, quick_streaming_tokens=0, **generation_kwargs):
        yield from self._infer_impl(spk_audio_prompt, text, output_path, None, emo_audio_prompt, emo_alpha, emo_vector, use_emo_text,
                                    emo_text, use_random, interval_silence, verbose, max_text_tokens_per_segment, stream_return, 1.0,
                                    True, generation_kwargs)

    def infer_stream(self, *a, **kw):
        raise NotImplementedError("IndexTTS-2 streaming needs the teacher-forced latent pass per chunk; the chunked path "
                                  "(infer_stream) is built for the v2.5 pipeline only")

    def _synthesize(self, segment_tokens: List[torch.Tensor], lang_ids, bundle, emovec, duration_factor, generation_kwargs,
                    max_text_tokens_per_segment) -> List[torch.Tensor]:
        gk = dict(generation_kwargs)
        gk.pop("do_sample", None)                       # popped and ignored by the reference (infer_v2.py:537,590)
        top_p, top_k = gk.pop("top_p", 0.8), gk.pop("top_k", 30)
        temperature = gk.pop("temperature", 0.8)
        length_penalty = gk.pop("length_penalty", 0.0)
        num_beams = gk.pop("num_beams", 3)
        repetition_penalty = gk.pop("repetition_penalty", 10.0)
        max_mel_tokens = gk.pop("max_mel_tokens", 1500)
        dev = self.device
        B = len(segment_tokens)
        L = max(int(t.numel()) for t in segment_tokens)
        text = torch.full((B, L), 1, dtype=torch.int32)                 # stop_text_token right padding
        # The teacher-forced pass sees `[start, ids, stop]` (infer_v2.py:558-560,639-642: the v2 reference tokenises a segment WITHOUT a
        # trailing stop id and `forward` pads one on).  The Frontend protocol appends stop id 1 to every segment (the v2.5 convention,
        # infer_v2_5.py:726); it is not part of the text, so the length handed to the latent pass excludes trailing stop ids.
        text_lens = []
        for i, t in enumerate(segment_tokens):
            flat = t.reshape(-1).to(torch.int32)
            text[i, : flat.numel()] = flat
            n = int(flat.numel())
            if n > 0 and int(flat[n - 1]) == 1:          # exactly the ONE stop id the Frontend protocol appends (ids inside a segment are >= 2)
                n -= 1
            if n <= 0:
                raise ValueError(f"segment {i} holds no text tokens (the reference never synthesises an empty segment, infer_v2.py:566-567)")
            text_lens.append(n)
        text_lens = torch.tensor(text_lens)
        spk_cond_emb, emo_cond_emb = bundle["spk_cond_emb"], bundle.get("emo_cond_emb", bundle["spk_cond_emb"])
        t0 = time.perf_counter()
        # one batch of B segments: the speaker latents are the same for every row (one speaker prompt).  The reference hands the
        # feature WIDTH over as the length (infer_v2.py:646, model_v2.py:761): "every frame valid" for prompts under 1024 frames
        n_spk = min(int(spk_cond_emb.shape[-1]), int(spk_cond_emb.shape[1]))
        lat1 = self.gpt.get_conditioning(spk_cond_emb.transpose(1, 2), torch.tensor([n_spk], device=spk_cond_emb.device))
        conds = self.gpt.conds_latent_v2(lat1.expand(B, -1, -1), emovec)
        codes, speech_conditioning_latent = self.gpt.inference_speech(
            spk_cond_emb, text.to(dev), emo_cond_emb, emo_vec=emovec, conds_latent=conds, do_sample=True, top_p=top_p, top_k=top_k,
            temperature=temperature, num_return_sequences=1, length_penalty=length_penalty, num_beams=num_beams,
            repetition_penalty=repetition_penalty, max_generate_length=max_mel_tokens, **gk)
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        t1 = time.perf_counter()
        if codes.shape[1] > 0 and (codes[:, -1] != self.stop_mel_token).any():
            warnings.warn(f"WARN: generation stopped due to exceeding `max_mel_tokens` ({max_mel_tokens}). "
                          f"Consider reducing `max_text_tokens_per_segment`({max_text_tokens_per_segment}) or increasing "
                          f"`max_mel_tokens`.", category=RuntimeWarning)
        codes, code_lens = self.trim_codes(codes)
        # teacher-forced latent pass (infer_v2.py:636-651) on the trimmed codes.  The reference runs it per segment at batch 1; the pass
        # has no attention mask (get_logits, model_v2.py:534), so a row padded to a longer neighbour's text / code length would see the
        # padding: rows are grouped by (text length, code length) and every group runs unpadded -- each row gets exactly its batch-1 result
        emo_b = emovec.expand(B, -1) if emovec.shape[0] == 1 else emovec
        cl = [int(v) for v in code_lens]
        groups = {}
        for b in range(B):
            groups.setdefault((int(text_lens[b]), cl[b]), []).append(b)
        latent = None
        for (tl, ml), rows in groups.items():
            if ml == 0:
                continue
            idx = torch.tensor(rows, device=codes.device)
            n = len(rows)
            lat_g = self.gpt(lat1.expand(n, -1, -1), text[rows][:, :tl].to(dev), torch.full((n,), tl), codes[idx][:, :ml],
                             torch.full((n,), ml), emo_cond_emb, cond_mel_lengths=None, emo_cond_mel_lengths=None, emo_vec=emo_b[rows],
                             use_speed=torch.zeros(n, dtype=torch.long))
            if latent is None:
                latent = torch.zeros(B, codes.shape[1], lat_g.shape[-1], dtype=torch.float32, device=codes.device)
            latent[idx, :ml] = lat_g.to(latent.device, torch.float32)
        if latent is None:                                     # every row stopped at once: nothing to render
            latent = torch.zeros(B, codes.shape[1], 1, dtype=torch.float32, device=codes.device)
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        t2 = time.perf_counter()
        mel, mel_lens = self.codes_latent_to_mel(codes, code_lens, latent, bundle)
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        t3 = time.perf_counter()
        wav = self.bigvgan(mel.float(), lens=mel_lens)
        wav = torch.clamp(PCM16_MAX * wav, -PCM16_MAX, PCM16_MAX)
        up = self.bigvgan.total_up
        out = [wav[i, :, : int(mel_lens[i]) * up].cpu() for i in range(wav.shape[0])]
        t4 = time.perf_counter()
        self.last_timing = dict(gpt=t1 - t0, gpt_forward=t2 - t1, s2mel=t3 - t2, bigvgan=t4 - t3)
        return out

    @torch.no_grad()
    def prompt_condition(self, spk_cond_emb: torch.Tensor, ref_mel_frames: int) -> torch.Tensor:
        """The speaker prompt's content condition of IndexTTS-2 (infer_v2.py:465-479) on the engine: `_, S_ref = semantic_codec.quantize(
        spk_cond_emb)` (down conv, Vocos encoder, FVQ search + out_project) -> `length_regulator(S_ref, ylens=[ref_mel.size(2)])`.
        spk_cond_emb (1, T, 1024) w2v-bert features -> (1, ref_mel_frames, 512).  (v2.5 regulates spk_cond_emb directly, infer_v2_5.py:651-656.)"""
        if self.s2mel is None or self.semantic_codec is None:
            raise RuntimeError("prompt_condition needs the engine's semantic codec and s2mel stages")
        _, S_ref = self.semantic_codec.quantize(spk_cond_emb)
        return self.s2mel.models["length_regulator"](S_ref, ylens=torch.tensor([int(ref_mel_frames)]), n_quantizers=3, f0=None)[0]

    def codes_latent_to_mel(self, codes, code_lens, latent, bundle, diffusion_steps: int = 25, inference_cfg_rate: float = 0.7,
                            noise=None):
        """infer_v2.py:653-676 for a batch of segments: gpt_layer(latent) + vq2emb(codes) -> length_regulator ->
        [prompt_condition | cond] -> cfm.inference -> drop the prompt frames, each row at its own lengths."""
        if self.s2mel is None or self.semantic_codec is None:
            return self.frontend.codes_latent_to_mel(codes, code_lens, latent, bundle)
        lens = [int(v) for v in code_lens]
        lat = self.s2mel.models["gpt_layer"](latent)                                         # (B, T, 1024)
        S_infer = self.semantic_codec.quantizer.vq2emb(codes.unsqueeze(1)).transpose(1, 2) + lat[:, : codes.shape[1]]
        target = [int(n * 1.72) for n in lens]                                               # (code_lens * 1.72).long()
        reg, cfm = self.s2mel.models["length_regulator"], self.s2mel.models["cfm"]
        cond = reg(S_infer, ylens=torch.tensor(target), n_quantizers=3, f0=None, xlens=lens, frame_lens=target)[0]
        prompt_condition, ref_mel, style = bundle["prompt_condition"], bundle["ref_mel"], bundle["style"]
        Tp, B = int(prompt_condition.shape[1]), codes.shape[0]
        total = [Tp + t for t in target]
        cat = torch.zeros(B, max(total), cond.shape[-1], dtype=torch.float32, device=cond.device)
        cat[:, :Tp] = prompt_condition.to(cond.device, torch.float32)
        for b in range(B):
            cat[b, Tp:total[b]] = cond[b, : target[b]]
        mel = cfm.inference(cat, torch.tensor(total), ref_mel, style, None, diffusion_steps, inference_cfg_rate=inference_cfg_rate,
                            noise=noise, frame_lens=total)
        return mel[:, :, Tp:].contiguous(), torch.tensor(target, dtype=torch.int32)
