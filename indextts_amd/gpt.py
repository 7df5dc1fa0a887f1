"""Host-side mirror of the reference `UnifiedVoice` decode interface, backed by the HIP engine.

Reference interface mirrored (indextts/gpt/model_v2.py): `UnifiedVoice(**cfg.gpt, spk_cond_mode="campplus")`,
`.load_state_dict`, `.post_init_gpt2_config(kv_cache=, half=)`, `.prepare_gpt_inputs(conds, text, langs)` (:648-714),
`.inference_speech(...)` (:716-825) returning `(codes[:, trunc_index:], speech_conditioning_latent)`, and the
teacher-forced `.forward(...)` latent pass (:596-646).  Input assembly (embedding gathers, left padding) is torch
plumbing on the device; the transformer stack, KV cache, logits processors, token selection and the per-token loop run
inside `libindextts_hip.so` (`itts_gpt_generate`), one hipGraph replay per token.

Prompt encoders (Conformer/Perceiver emotion encoder, CAMPPlus) are out of this path: pass
`emo_vec=` / `campplus_embedding=` computed by the PyTorch-ROCm modules, as `indextts/infer_v2_5.py:762-781` does.
"""
import ctypes as C
from typing import List, Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from . import _lib


# HF `generate` options that change the generated ids and that the device loop does not implement (the reference forwards every
# hf_generate_kwarg to `GenerationMixin.generate`, model_v2.py:815-820): passing one raises instead of being ignored.
_UNSUPPORTED_GENERATE_KWARGS = frozenset((
    "no_repeat_ngram_size", "encoder_no_repeat_ngram_size", "bad_words_ids", "min_length", "min_new_tokens", "num_beam_groups",
    "diversity_penalty", "penalty_alpha", "early_stopping", "stopping_criteria", "prefix_allowed_tokens_fn", "constraints",
    "force_words_ids", "suppress_tokens", "begin_suppress_tokens", "forced_bos_token_id", "forced_eos_token_id",
    "encoder_repetition_penalty", "epsilon_cutoff", "eta_cutoff", "exponential_decay_length_penalty", "renormalize_logits",
    "guidance_scale", "sequence_bias", "min_p", "typical_p", "assistant_model", "negative_prompt_ids"))


def _reject_logits_processor(hf_generate_kwargs: dict):
    """The reference builds its own `logits_processor` list and passes it beside **hf_generate_kwargs (model_v2.py:793-799,815-820;
    model.py:655-660), so a caller-supplied one is a TypeError there (duplicate keyword); the same error here, not a silent drop."""
    if "logits_processor" in hf_generate_kwargs:
        raise TypeError("inference_speech() got multiple values for keyword argument 'logits_processor' (the reference passes its own "
                        "LogitsProcessorList to generate; use typical_sampling / typical_mass)")


class UnifiedVoice:
    """`spk_cond_mode="campplus"` (IndexTTS-2.5, infer_v2_5.py:139): 3 conditioning tokens from the CAMPPlus style vector.
    Any other mode (IndexTTS-2, infer_v2.py:98; the reference default is "conformer"): 34 conditioning tokens -- 32 latents of
    the Conformer + Perceiver speaker encoder (`get_conditioning`, a prompt-side PyTorch module: set `conditioning_fn` or pass
    `conds_latent=`) plus the two `speed_emb` rows (model_v2.py:767-773); no language embedding (:680)."""
    _ALLOW_ENCODER_CONDITIONING = True

    def __init__(self, layers=8, model_dim=512, heads=8, max_text_tokens=120, max_mel_tokens=250,
                 max_conditioning_inputs=1, mel_length_compression=1024, number_text_tokens=256, start_text_token=0,
                 stop_text_token=1, number_mel_codes=8194, start_mel_token=8192, stop_mel_token=8193,
                 train_solo_embeddings=False, use_mel_codes_as_input=True, checkpointing=True, types=1,
                 condition_num_latent=32, condition_type="perceiver", condition_module=None, emo_condition_module=None,
                 use_accel=False, spk_cond_mode="conformer", precision="bf16", device="cuda:0", conditioning_fn=None, **_unused):
        self.conditioning_fn = conditioning_fn
        self.cond_num = condition_num_latent
        # Conformer + Perceiver conditioning encoders on the engine (indextts_amd/cond.py): built in load_state_dict when the config
        # carries their `condition_module` / `emo_condition_module` sections and the checkpoint their weights
        self.condition_type, self.condition_module, self.emo_condition_module = condition_type, condition_module, emo_condition_module
        self.cond_encoders = None
        self.layers, self.model_dim, self.heads = layers, model_dim, heads
        self.max_text_tokens, self.max_mel_tokens = max_text_tokens, max_mel_tokens
        self.max_conditioning_inputs = max_conditioning_inputs
        self.number_text_tokens, self.number_mel_codes = number_text_tokens, number_mel_codes
        self.start_text_token, self.stop_text_token = start_text_token, stop_text_token
        self.start_mel_token, self.stop_mel_token = start_mel_token, stop_mel_token
        self.types = types
        self.mel_length_compression = mel_length_compression
        self.spk_cond_mode = spk_cond_mode
        self.device = torch.device(device)
        self.precision = {"bf16": 1, "bfloat16": 1, "fp32": 0, "float32": 0, "f32": 0}[precision]
        self.kv_cache = True
        self.use_graph = True
        self.n_mel_pos = max_mel_tokens + 2 + max_conditioning_inputs
        self.n_text_pos = max_text_tokens + 2
        # `tts.gpt.text_pos_embedding.emb.num_embeddings` is read by the reference's text splitter (indextts/infer_v2_5.py:428) and
        # `mel_pos_embedding` likewise by callers sizing generation: attribute paths kept
        from types import SimpleNamespace
        self.text_pos_embedding = SimpleNamespace(emb=SimpleNamespace(num_embeddings=self.n_text_pos))
        self.mel_pos_embedding = SimpleNamespace(emb=SimpleNamespace(num_embeddings=self.n_mel_pos))
        cfg = _lib.GPTConfig()
        cfg.layers, cfg.model_dim, cfg.heads = layers, model_dim, heads
        cfg.vocab, cfg.n_mel_pos, cfg.precision = number_mel_codes, self.n_mel_pos, self.precision
        cfg.start_mel_token, cfg.stop_mel_token, cfg.ln_eps = start_mel_token, stop_mel_token, 1e-5
        self._h = C.c_void_p()
        with _lib.on_device(self.device):          # the handle (weights, stream, events) is bound to the device current here
            _lib.check(_lib.lib().itts_gpt_create(C.byref(cfg), C.byref(self._h)), "itts_gpt_create")
        self._emb: Dict[str, torch.Tensor] = {}
        self._loaded = False
        self._ws = None
        self._bufs = {}
        self.last_timing = None

    # ---- checkpoint ------------------------------------------------------------------------------------------
    _ENGINE_PREFIXES = ("gpt.h.", "gpt.ln_f.", "final_norm.", "mel_head.")
    _HOST_TENSORS = ("mel_embedding.weight", "mel_pos_embedding.emb.weight", "text_embedding.weight",
                     "text_pos_embedding.emb.weight", "lang_embedding.weight", "spk_emb_proj.weight",
                     "spk_emb_proj.bias")

    _OPTIONAL_HOST_TENSORS = ("lang_embedding.weight",)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        """Reference `gpt.pth` names (`strict=False` like indextts/utils/checkpoint.py:22-29); returns ignored keys."""
        L = _lib.lib()
        ignored = []
        for name, t in sd.items():
            if name in self._HOST_TENSORS:
                self._emb[name] = t.detach().to(self.device, torch.float32).contiguous()
            if name.startswith(self._ENGINE_PREFIXES) or name in ("mel_embedding.weight", "mel_pos_embedding.emb.weight"):
                if name.endswith(".attn.bias") or name.endswith(".attn.masked_bias"):
                    ignored.append(name)      # HF causal-mask buffers
                    continue
                tt = t.detach().to("cpu", torch.float32).contiguous()
                shape = (C.c_int64 * tt.dim())(*tt.shape)
                _lib.check(L.itts_gpt_load_tensor(self._h, name.encode(), C.c_void_p(tt.data_ptr()), shape, tt.dim()),
                           f"itts_gpt_load_tensor({name})")
            elif name not in self._HOST_TENSORS:
                ignored.append(name)
        _lib.check(L.itts_gpt_finalize(self._h), "itts_gpt_finalize")
        if "speed_emb.weight" in sd:
            self._emb["speed_emb.weight"] = sd["speed_emb.weight"].detach().to(self.device, torch.float32).contiguous()
        optional = set(self._OPTIONAL_HOST_TENSORS)
        if getattr(self, "spk_cond_mode", "campplus") != "campplus":          # IndexTTS-2: no style projection, speed embedding instead
            optional |= {"spk_emb_proj.weight", "spk_emb_proj.bias"}
            if type(self) is UnifiedVoice and "speed_emb.weight" not in self._emb:
                raise _lib.HipEngineError("UnifiedVoice.load_state_dict: missing ['speed_emb.weight']")
        missing = [n for n in self._HOST_TENSORS if n not in self._emb and n not in optional]
        if missing:
            raise _lib.HipEngineError(f"UnifiedVoice.load_state_dict: missing {missing}")
        emo_cm = getattr(self, "emo_condition_module", None)
        if emo_cm is not None and "emo_conditioning_encoder.after_norm.weight" in sd and "emovec_layer.weight" in sd:
            from .cond import ConditioningEncoders
            spk_cm = self.condition_module if (self.condition_type == "conformer_perceiver" and self.spk_cond_mode != "campplus"
                                                and "conditioning_encoder.after_norm.weight" in sd) else None
            self.cond_encoders = ConditioningEncoders(self.model_dim, spk_cm, emo_cm, cond_num=self.cond_num, device=self.device)
            self.cond_encoders.load_state_dict(sd)
            ignored = [n for n in ignored if not n.startswith(("emo_conditioning_encoder.", "emo_perceiver_encoder.", "emovec_layer.", "emo_layer."))
                       and not (spk_cm is not None and n.startswith(("conditioning_encoder.", "perceiver_encoder.")))]
        self._loaded = True
        return ignored

    def post_init_gpt2_config(self, use_deepspeed=False, kv_cache=False, half=False):
        """model_v2.py:422-493.  kv_cache=False reproduces the reference's no-cache position rule (positions 0..n-1)
        -- the engine always keeps a KV cache; only the mel position index changes."""
        self.kv_cache = bool(kv_cache)
        return self

    def eval(self):
        return self

    def to(self, device):
        """The engine lives on the device given at construction (`device=`); moving a loaded engine is not supported."""
        d = torch.device(device)
        if d.type == "cuda" and d.index is not None and self.device.index is not None and d.index != self.device.index:
            raise _lib.HipEngineError(f"UnifiedVoice was built on {self.device}; construct it with device={device!r} instead")
        return self

    # ---- input assembly (model_v2.py:648-714) ------------------------------------------------------------------
    def prepare_gpt_inputs(self, conditional_latents: torch.Tensor, text_inputs: torch.Tensor,
                           langs: Optional[torch.Tensor] = None):
        dev = self.device
        text_inputs = text_inputs.to(dev)
        conditional_latents = conditional_latents.to(dev, torch.float32)
        b, L = text_inputs.shape[:2]
        single_cond = conditional_latents.ndim == 3 and conditional_latents.shape[0] == 1
        if not single_cond:
            assert conditional_latents.shape[0] == b, f"batch size mismatch: {conditional_latents.shape[0]} vs {b}"
        n_cond = conditional_latents.shape[1]
        target_len = n_cond + L + 2
        valid = (text_inputs != self.stop_text_token) & (text_inputs != self.start_text_token)
        n_valid = valid.sum(dim=1)                                     # (b,)
        padding = L - n_valid                                           # left pad per row
        if L + 2 > self.n_text_pos and int(n_valid.max()) + 2 > self.n_text_pos:
            # the reference indexes text_pos_embedding out of range here (IndexError on CPU, device assert on a GPU)
            raise ValueError(f"text of {int(n_valid.max())} tokens exceeds max_text_tokens={self.max_text_tokens}")
        # gather valid ids to the right end of an (b, L+2) row: [pad..][start][ids][stop]
        order = torch.argsort((~valid).to(torch.int8), dim=1, stable=True)          # valid ids first, in order
        ids_sorted = torch.gather(text_inputs.long(), 1, order)
        rows = torch.full((b, L + 2), self.stop_text_token, dtype=torch.long, device=dev)
        ar = torch.arange(L + 2, device=dev)[None, :]
        start_col = padding[:, None]
        rel = ar - start_col                                            # position within [start, ids..., stop]
        in_ids = (rel >= 1) & (rel <= n_valid[:, None])
        src = (rel - 1).clamp(0, L - 1)
        rows = torch.where(in_ids, torch.gather(ids_sorted, 1, src), rows)
        rows = torch.where(rel == 0, torch.full_like(rows, self.start_text_token), rows)
        tok_valid = rel >= 0
        pos = rel.clamp(min=0)
        emb = self._emb["text_embedding.weight"][rows] + self._emb["text_pos_embedding.emb.weight"][pos]
        if langs is not None and "lang_embedding.weight" in self._emb and getattr(self, "spk_cond_mode", "campplus") == "campplus":
            lg = langs.to(dev).long().reshape(-1)
            if lg.numel() == 1:
                lg = lg.expand(b)
            emb = emb + self._emb["lang_embedding.weight"][lg][:, None, :]
        emb = emb * tok_valid[..., None]                                # zero rows on the left pad
        cond = conditional_latents.expand(b, -1, -1) if single_cond else conditional_latents
        # [pad][cond][text]: roll the cond block to sit right after the pad
        out = torch.zeros(b, target_len, self.model_dim, dtype=torch.float32, device=dev)
        col = torch.arange(target_len, device=dev)[None, :]
        is_cond = (col >= padding[:, None]) & (col < padding[:, None] + n_cond)
        cond_idx = (col - padding[:, None]).clamp(0, n_cond - 1)
        out = torch.where(is_cond[..., None], torch.gather(cond, 1, cond_idx[..., None].expand(-1, -1, self.model_dim)), out)
        is_text = col >= padding[:, None] + n_cond
        text_idx = (col - n_cond).clamp(0, L + 1).expand(b, -1)
        out = torch.where(is_text[..., None], torch.gather(emb, 1, text_idx[..., None].expand(-1, -1, self.model_dim)), out)
        attention_mask = torch.ones(b, target_len + 1, dtype=torch.long, device=dev)
        attention_mask[:, :target_len] = (col >= padding[:, None]).long()
        fake_inputs = torch.ones(b, target_len + 1, dtype=torch.long, device=dev)
        fake_inputs[:, -1] = self.start_mel_token
        return fake_inputs, out, attention_mask

    def _spk_proj_params(self):
        """float64 host copies of spk_emb_proj's weight / bias, made once per loaded state dict (`_spk_proj`)"""
        w = self._emb["spk_emb_proj.weight"]
        hit = getattr(self, "_spk64", None)
        if hit is None or hit[0] is not w:
            hit = (w, w.detach().to("cpu", torch.float64), self._emb["spk_emb_proj.bias"].detach().to("cpu", torch.float64))
            self._spk64 = hit
        return hit[1], hit[2]

    def conds_latent(self, campplus_embedding: torch.Tensor, emo_vec: torch.Tensor) -> torch.Tensor:
        """spk_emb_proj(style) + emo_vec, then two zero tokens (model_v2.py:754-755,768)."""
        dev = self.device
        spk = _spk_proj(campplus_embedding.to(dev, torch.float32), *self._spk_proj_params())
        spk = spk.unsqueeze(0) if spk.ndim != 3 else spk
        emo_vec = emo_vec.to(dev, torch.float32)
        return torch.cat((spk + emo_vec.unsqueeze(1), torch.zeros(spk.size(0), 2, spk.size(2), device=dev)), 1), spk

    def get_conditioning(self, speech_conditioning_input, cond_mel_lengths=None):
        """IndexTTS-2 speaker latents (model_v2.py:556-586): Conformer + Perceiver -- on the engine when the checkpoint carried the
        encoders (`cond_encoders`), else through the injected `conditioning_fn` (e.g. the reference module's bound method)."""
        if self.conditioning_fn is not None:
            return self.conditioning_fn(speech_conditioning_input, cond_mel_lengths)
        if self.cond_encoders is not None and self.cond_encoders.spk is not None:
            if cond_mel_lengths is None:
                cond_mel_lengths = torch.full((speech_conditioning_input.shape[0],), speech_conditioning_input.shape[-1])
            return self.cond_encoders.get_conditioning(speech_conditioning_input, cond_mel_lengths)
        raise NotImplementedError("no conditioning encoder: load a checkpoint with conditioning_encoder.* / perceiver_encoder.* and a "
                                  "`condition_module` config, set UnifiedVoice.conditioning_fn, or pass conds_latent=")

    def get_emo_conditioning(self, speech_conditioning_input, cond_mel_lengths=None):        # model_v2.py:588-593
        self._need_cond("get_emo_conditioning")
        if cond_mel_lengths is None:
            cond_mel_lengths = torch.full((speech_conditioning_input.shape[0],), speech_conditioning_input.shape[-1])
        return self.cond_encoders.get_emo_conditioning(speech_conditioning_input, cond_mel_lengths)

    def get_emovec(self, emo_speech_conditioning_latent, emo_cond_lengths):                   # model_v2.py:827-831
        self._need_cond("get_emovec")
        return self.cond_encoders.get_emovec(emo_speech_conditioning_latent, emo_cond_lengths)

    def merge_emovec(self, speech_conditioning_latent, emo_speech_conditioning_latent, cond_lengths, emo_cond_lengths, alpha=1.0):
        self._need_cond("merge_emovec")                                                       # model_v2.py:833-838
        return self.cond_encoders.merge_emovec(speech_conditioning_latent, emo_speech_conditioning_latent, cond_lengths, emo_cond_lengths, alpha)

    def _need_cond(self, who):
        if self.cond_encoders is None:
            raise NotImplementedError(f"{who}: the checkpoint / config carried no emotion Conformer + Perceiver encoders "
                                      "(emo_condition_module, emo_conditioning_encoder.*, emo_perceiver_encoder.*, emovec_layer.*, emo_layer.*)")

    def conds_latent_v2(self, speech_conditioning_latent: torch.Tensor, emo_vec: torch.Tensor) -> torch.Tensor:
        """34 conditioning tokens of IndexTTS-2 (model_v2.py:767-773): latents + emo_vec, speed_emb(1), speed_emb(0)."""
        dev = self.device
        lat = speech_conditioning_latent.to(dev, torch.float32)
        se = self._emb["speed_emb.weight"]
        b = lat.shape[0]
        return torch.cat((lat + emo_vec.to(dev, torch.float32).unsqueeze(1), se[1].expand(b, 1, -1), se[0].expand(b, 1, -1)), 1)

    # ---- generation ----------------------------------------------------------------------------------------------
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def _persistent(self, name: str, shape, dtype) -> torch.Tensor:
        key = (name, tuple(int(v) for v in shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            if len(self._bufs) >= 32:
                self._bufs.clear()
            t = self._bufs[key] = torch.empty(*key[1], dtype=dtype, device=self.device)
        return t

    @staticmethod
    def _seed(seed, do_sample, uniforms) -> int:
        """Device RNG seed.  `seed=None` (the default) draws it from torch's global generator, so sampled calls differ from
        call to call and follow `torch.manual_seed` like the reference's `torch.multinomial` does; an explicit seed is kept."""
        if seed is not None:
            return int(seed)
        if not do_sample or uniforms is not None:
            return 0
        return int(torch.randint(0, 2 ** 62, (1,)).item())

    def set_compaction(self, enable: bool = True, granularity: int = 8):
        """Row compaction of ragged decode batches (itts_gpt_set_compaction): finished rows leave the running batch in buckets of
        `granularity` rows.  On by default; results do not depend on it."""
        _lib.check(_lib.lib().itts_gpt_set_compaction(self._h, int(bool(enable)), int(granularity)), "itts_gpt_set_compaction")

    def graph_stats(self) -> dict:
        """decode-step hipGraphs captured / reused by this engine handle (itts_gpt_graph_stats)"""
        cap, hit = C.c_int32(0), C.c_int32(0)
        _lib.lib().itts_gpt_graph_stats(self._h, C.byref(cap), C.byref(hit))
        return dict(captures=cap.value, hits=hit.value)

    def generate(self, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor, max_new_tokens: int, do_sample=False,
                 num_beams=1, top_p=1.0, top_k=50, temperature=1.0, repetition_penalty=1.0, length_penalty=1.0,
                 uniforms: Optional[torch.Tensor] = None, seed: Optional[int] = None, typical_mass: float = 0.0,
                 row_max_new: Optional[Sequence[int]] = None, **unused) -> torch.Tensor:
        """`row_max_new` (engine extension for merged batches): per-row cap on generated tokens -- row b emits the stop token from token
        index row_max_new[b] on, i.e. each request of a batch keeps its own `max_mel_tokens` (sampling / greedy only).
        `GPT2InferenceModel.generate` for greedy / multinomial sampling (typical_mass > 0: the reference's
        TypicalLogitsWarper sits between the repetition penalty and the warpers, model_v2.py:794-799).  inputs_embeds (B,s,D) = the cached prefix;
        attention_mask (B,s+1).  Returns generated ids (B, n) (what `output[:, trunc_index:]` is in the reference)."""
        if not self._loaded:
            raise RuntimeError("UnifiedVoice: load_state_dict() first")
        altering = sorted(k for k in unused if k in _UNSUPPORTED_GENERATE_KWARGS and unused[k] is not None)
        if altering:             # the reference forwards these to HF `generate`, where they change the ids: never drop them silently
            raise NotImplementedError(f"generate: {altering} would change the generated ids and the device loop does not implement them "
                                      "(supported: do_sample, num_beams, top_p, top_k, temperature, repetition_penalty, length_penalty, "
                                      "typical sampling)")
        self._check_idle("generate")
        if num_beams != 1:
            if row_max_new is not None:
                raise NotImplementedError("generate: row_max_new (per-row token caps of a merged batch) is implemented for num_beams=1 only; "
                                          "the beam kernels take one max_new_tokens per call")
            return self._generate_beam(inputs_embeds, attention_mask, max_new_tokens, do_sample, num_beams, top_p, top_k,
                                       temperature, repetition_penalty, length_penalty, uniforms, seed, typical_mass)
        dev = self.device
        B, s, D = inputs_embeds.shape
        start = (self._emb["mel_embedding.weight"][self.start_mel_token] + self._emb["mel_pos_embedding.emb.weight"][0])
        x = torch.cat([inputs_embeds.to(dev, torch.float32), start.expand(B, 1, D)], dim=1).contiguous()
        S = s + 1
        pad = (attention_mask[:, :S] == 0).sum(dim=1).to(torch.int32).contiguous()
        gp = _lib.GenParams()
        gp.do_sample, gp.num_beams, gp.top_k = int(bool(do_sample)), 1, int(top_k or 0)
        gp.min_tokens_to_keep, gp.max_new_tokens = 1, int(max_new_tokens)
        gp.pos_offset = 2 if self.kv_cache else 1
        gp.top_p, gp.temperature = float(top_p), float(temperature)
        gp.repetition_penalty = float(repetition_penalty if repetition_penalty is not None else 1.0)
        gp.length_penalty, gp.seed = float(length_penalty), self._seed(seed, do_sample, uniforms)
        gp.typical_mass = float(typical_mass)
        L = _lib.lib()
        Tmax = S + int(max_new_tokens)
        need = L.itts_gpt_workspace_bytes(self._h, B, S, Tmax)
        ws = self._workspace(need)
        # output / uniforms buffers persist per shape: their addresses are part of the cached decode graph's key
        codes = self._persistent("codes", (B, int(max_new_tokens)), torch.int64)
        n_steps = C.c_int32(0)
        pen = (C.c_int32 * 2)(1, self.start_mel_token)          # fake prefix ids (all ones) + start_mel
        u = None
        if uniforms is not None:
            if uniforms.shape[0] < max_new_tokens or uniforms.shape[1] != B:
                raise ValueError("uniforms must be (>= max_new_tokens, B)")
            u = self._persistent("uniforms", (int(max_new_tokens), B), torch.float64)
            u.copy_(uniforms[: int(max_new_tokens)])
        lim = None
        if row_max_new is not None:
            if len(row_max_new) != B:
                raise ValueError(f"row_max_new must have one entry per row ({B}), got {len(row_max_new)}")
            lim = self._persistent("row_limits", (B,), torch.int32)           # persistent: its address is part of the decode graph's key
            lim.copy_(torch.as_tensor([int(v) for v in row_max_new], dtype=torch.int32))
        _lib.check(L.itts_gpt_set_row_limits(self._h, _lib.ptr(lim), B if lim is not None else 0), "itts_gpt_set_row_limits")
        try:
            rc = L.itts_gpt_generate(self._h, _lib.ptr(x), _lib.ptr(pad), B, S, C.byref(gp), pen, 2, _lib.ptr(u),
                                     _lib.ptr(codes), C.byref(n_steps), _lib.ptr(ws), ws.numel(), int(self.use_graph),
                                     _lib.stream_ptr(self.device))
        finally:
            if lim is not None:
                L.itts_gpt_set_row_limits(self._h, None, 0)
        _lib.check(rc, "itts_gpt_generate")
        pm, dm, st = C.c_float(0), C.c_float(0), C.c_int32(0)
        L.itts_gpt_last_timing(self._h, C.byref(pm), C.byref(dm), C.byref(st))
        rs, nc = C.c_int64(0), C.c_int32(0)
        L.itts_gpt_compaction_stats(self._h, C.byref(rs), C.byref(nc))
        # row_steps: sum over the decode steps of the rows each step ran (B x steps when no row left the batch early)
        self.last_timing = dict(prefill_ms=pm.value, decode_ms=dm.value, steps=st.value, row_steps=int(rs.value), compactions=int(nc.value))
        # HF stops right after the step at which every row has emitted EOS
        is_stop = codes == self.stop_mel_token
        first = torch.where(is_stop.any(1), is_stop.int().argmax(1) + 1, torch.full((B,), codes.shape[1], device=dev))
        n = int(min(int(first.max().item()), n_steps.value))
        return codes[:, :n].clone()

    def generate_chunks(self, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor, max_new_tokens: int, chunk_size: int,
                        overlap_size: int, do_sample=False, num_beams=1, top_p=1.0, top_k=50, temperature=1.0, repetition_penalty=1.0,
                        length_penalty=1.0, uniforms: Optional[torch.Tensor] = None, seed: Optional[int] = None,
                        typical_mass: float = 0.0, **unused):
        """Streaming form of `generate` (`GPTTRTEngine.generate_chunks`, backends/trt/runtime/gpt_trtllm_runtime.py:381-520):
        yields `(chunk_codes (B, <= chunk_size), is_last, batch_done [B], chunk_code_lens (B,))` as soon as `chunk_size` codes
        exist, consecutive chunks overlapping by `overlap_size` codes; the decode loop is suspended between chunks with its whole
        state (KV cache, position, finished flags) on the device (`itts_gpt_generate_chunk`).  Sampling / greedy only: the engine
        streams one hypothesis per row."""
        stride = int(chunk_size) - int(overlap_size)
        if stride <= 0:
            raise ValueError(f"overlap_size ({overlap_size}) must be less than chunk_size ({chunk_size}); "
                             f"got stride={stride} which would cause an infinite loop.")
        if num_beams != 1:
            raise NotImplementedError("generate_chunks streams num_beams=1 (a beam's prefix is not final until the search ends)")
        if not self._loaded:
            raise RuntimeError("UnifiedVoice: load_state_dict() first")
        # the suspended loop's state (KV cache in the workspace, the persistent `codes` buffer) is shared with generate(): another
        # generation on this object while a stream is open would corrupt it -> refuse until the stream is exhausted / closed
        self._check_idle("generate_chunks")
        self._stream_open = True
        try:
            yield from self._generate_chunks_body(inputs_embeds, attention_mask, max_new_tokens, chunk_size, overlap_size, stride, do_sample,
                                                  top_p, top_k, temperature, repetition_penalty, length_penalty, uniforms, seed, typical_mass)
        finally:
            self._stream_open = False

    def _check_idle(self, who: str):
        if getattr(self, "_stream_open", False):
            raise RuntimeError(f"UnifiedVoice.{who}: a chunked generation (generate_chunks / infer_stream) is open on this engine; its "
                               "KV cache and code buffer live in the shared workspace.  Exhaust or close() that generator first, or use "
                               "a second UnifiedVoice for concurrent requests.")

    def _generate_chunks_body(self, inputs_embeds, attention_mask, max_new_tokens, chunk_size, overlap_size, stride, do_sample, top_p, top_k,
                              temperature, repetition_penalty, length_penalty, uniforms, seed, typical_mass):
        dev = self.device
        B, s, D = inputs_embeds.shape
        start = (self._emb["mel_embedding.weight"][self.start_mel_token] + self._emb["mel_pos_embedding.emb.weight"][0])
        x = torch.cat([inputs_embeds.to(dev, torch.float32), start.expand(B, 1, D)], dim=1).contiguous()
        S, max_new = s + 1, int(max_new_tokens)
        pad = (attention_mask[:, :S] == 0).sum(dim=1).to(torch.int32).contiguous()
        gp = _lib.GenParams()
        gp.do_sample, gp.num_beams, gp.top_k = int(bool(do_sample)), 1, int(top_k or 0)
        gp.min_tokens_to_keep, gp.max_new_tokens = 1, max_new
        gp.pos_offset = 2 if self.kv_cache else 1
        gp.top_p, gp.temperature = float(top_p), float(temperature)
        gp.repetition_penalty = float(repetition_penalty if repetition_penalty is not None else 1.0)
        gp.length_penalty, gp.seed = float(length_penalty), self._seed(seed, do_sample, uniforms)
        gp.typical_mass = float(typical_mass)
        L = _lib.lib()
        ws = self._workspace(L.itts_gpt_workspace_bytes(self._h, B, S, S + max_new))
        codes = self._persistent("codes", (B, max_new), torch.int64)
        u = None
        if uniforms is not None:
            if uniforms.shape[0] < max_new or uniforms.shape[1] != B:
                raise ValueError("uniforms must be (>= max_new_tokens, B)")
            u = self._persistent("uniforms", (max_new, B), torch.float64)
            u.copy_(uniforms[:max_new])
        pen = (C.c_int32 * 2)(1, self.start_mel_token)
        n_steps = C.c_int32(0)
        next_chunk_at = int(chunk_size)
        first = True
        while True:
            limit = min(next_chunk_at, max_new)
            rc = L.itts_gpt_generate_chunk(self._h, _lib.ptr(x) if first else None, _lib.ptr(pad), B, S, C.byref(gp), pen, 2, _lib.ptr(u),
                                           _lib.ptr(codes), limit, C.byref(n_steps), _lib.ptr(ws), ws.numel(), int(self.use_graph),
                                           _lib.stream_ptr(self.device))
            _lib.check(rc, "itts_gpt_generate_chunk")
            first = False
            steps = int(n_steps.value)
            got = codes[:, :steps]
            is_stop = got == self.stop_mel_token
            done = is_stop.any(1)
            lens = torch.where(done, is_stop.int().argmax(1), torch.full((B,), steps, device=dev))      # codes before the stop token
            current_len = int(lens.max().item())
            done_l, lens_l = done.tolist(), lens.tolist()
            finished = all(done_l) or steps >= max_new or steps < limit
            while current_len >= next_chunk_at:
                pos = next_chunk_at - chunk_size
                yield (codes[:, pos:next_chunk_at].clone(), False, [d and n <= next_chunk_at for d, n in zip(done_l, lens_l)],
                       torch.tensor([max(0, min(n - pos, chunk_size)) for n in lens_l], dtype=torch.long, device=dev))
                next_chunk_at += stride
            if finished:
                pos = next_chunk_at - chunk_size
                if pos < current_len:
                    yield (codes[:, pos:current_len].clone(), True, [True] * B,
                           torch.tensor([max(0, n - pos) for n in lens_l], dtype=torch.long, device=dev))
                return

    # ---- beam search / beam-sample (num_beams > 1; the reference default is 3-beam beam-sample) ------------------------
    def _generate_beam(self, inputs_embeds, attention_mask, max_new_tokens, do_sample, num_beams, top_p, top_k, temperature,
                       repetition_penalty, length_penalty, uniforms, seed, typical_mass=0.0) -> torch.Tensor:
        dev = self.device
        nb = int(num_beams)
        B, s, D = inputs_embeds.shape
        start = (self._emb["mel_embedding.weight"][self.start_mel_token] + self._emb["mel_pos_embedding.emb.weight"][0])
        x = torch.cat([inputs_embeds.to(dev, torch.float32), start.expand(B, 1, D)], dim=1)
        x = x.repeat_interleave(nb, dim=0).contiguous()               # _expand_inputs_for_generation: beams adjacent
        S = s + 1
        pad = (attention_mask[:, :S] == 0).sum(dim=1).to(torch.int32).repeat_interleave(nb).contiguous()
        nseq, max_new = B * nb, int(max_new_tokens)
        gp = _lib.GenParams()
        gp.do_sample, gp.num_beams, gp.top_k = int(bool(do_sample)), nb, int(top_k or 0)
        gp.min_tokens_to_keep, gp.max_new_tokens = 2, max_new        # one eos id -> keep eos + 1 (generation_utils.py:1023-1029)
        gp.pos_offset = 2 if self.kv_cache else 1
        gp.top_p, gp.temperature = float(top_p), float(temperature)
        gp.repetition_penalty = float(repetition_penalty if repetition_penalty is not None else 1.0)
        gp.length_penalty, gp.seed = float(length_penalty), self._seed(seed, do_sample, uniforms)
        gp.typical_mass = float(typical_mass)
        L = _lib.lib()
        Tmax = S + max_new
        ws = self._workspace(L.itts_gpt_beam_workspace_bytes(self._h, B, nb, S, Tmax))
        hist_tok = torch.empty(max_new, nseq, dtype=torch.int32, device=dev)
        hist_par = torch.empty(max_new, nseq, dtype=torch.int32, device=dev)
        beam_scores = torch.empty(nseq, dtype=torch.float32, device=dev)
        hyps = torch.empty(B, 4, 4, dtype=torch.float32, device=dev)          # {f32 score, i32 step, i32 row, pad}
        n_hyps = torch.empty(B, dtype=torch.int32, device=dev)
        done = torch.empty(B, dtype=torch.uint8, device=dev)
        n_steps = C.c_int32(0)
        pen = (C.c_int32 * 2)(1, self.start_mel_token)
        u = None
        if uniforms is not None:
            if uniforms.dim() != 3 or uniforms.shape[0] < max_new or uniforms.shape[1] != B or uniforms.shape[2] != 2 * nb:
                raise ValueError("uniforms must be (>= max_new_tokens, B, 2*num_beams)")
            u = self._persistent("beam_uniforms", (max_new, B, 2 * nb), torch.float64)
            u.copy_(uniforms[:max_new])
        rc = L.itts_gpt_generate_beam(self._h, _lib.ptr(x), _lib.ptr(pad), B, nb, S, C.byref(gp), pen, 2, _lib.ptr(u),
                                      _lib.ptr(hist_tok), _lib.ptr(hist_par), _lib.ptr(beam_scores), _lib.ptr(hyps),
                                      _lib.ptr(n_hyps), _lib.ptr(done), C.byref(n_steps), _lib.ptr(ws), ws.numel(),
                                      int(self.use_graph), _lib.stream_ptr(self.device))
        _lib.check(rc, "itts_gpt_generate_beam")
        pm, dm, st = C.c_float(0), C.c_float(0), C.c_int32(0)
        L.itts_gpt_last_timing(self._h, C.byref(pm), C.byref(dm), C.byref(st))
        self.last_timing = dict(prefill_ms=pm.value, decode_ms=dm.value, steps=st.value)
        # ---- BeamSearchScorer.finalize (transformers_beam_search.py:320-408) on the host ----
        ht, hp = hist_tok.cpu().numpy(), hist_par.cpu().numpy()
        bs = beam_scores.cpu().numpy()
        hy_f = hyps.cpu()
        hy_i = hy_f.view(torch.int32).numpy()
        hy_s = hy_f.numpy()
        nh, dn = n_hyps.cpu().numpy(), done.cpu().numpy()
        steps_run = int(n_steps.value)
        # the reference loop ends at the first step after which every utterance is done (or at max_length)
        if dn.all():
            last = max(int(hy_i[b, q, 1]) for b in range(B) for q in range(int(nh[b])))
            steps_run = min(steps_run, last + 1)

        def seq_of(row: int, upto: int):
            toks = []
            r = row
            for sidx in range(upto, -1, -1):
                toks.append(int(ht[sidx, r]))
                r = int(hp[sidx, r])
            return toks[::-1]

        stop = self.stop_mel_token
        best = []
        for b in range(B):
            heap = [(float(hy_s[b, q, 0]), seq_of(int(hy_i[b, q, 2]), int(hy_i[b, q, 1]) - 1) if int(hy_i[b, q, 1]) > 0 else [])
                    for q in range(int(nh[b]))]
            if not dn[b]:
                worst = min([h0[0] for h0 in heap], default=1e9) if len(heap) >= nb else 1e9
                for j in range(nb):                         # open beams join the heap with generated_len = steps
                    row = b * nb + j
                    sc = float(bs[row]) / (steps_run ** float(length_penalty))
                    if len(heap) < nb or sc > worst:
                        heap.append((sc, seq_of(row, steps_run - 1)))
                        if len(heap) > nb:
                            heap.remove(min(heap, key=lambda t: t[0]))
                        worst = min(t[0] for t in heap)
            best.append(sorted(heap, key=lambda t: t[0])[-1][1])
        lens = [len(t) for t in best]
        sent_max = min(max(lens) + 1, max_new)
        out = torch.full((B, sent_max), stop, dtype=torch.int64)
        for b, t in enumerate(best):
            out[b, : len(t)] = torch.tensor(t[:sent_max], dtype=torch.int64)
        return out.to(dev)

    def _prepare_inference(self, speech_condition, text_inputs, langs, cond_lengths, emo_vec, campplus_embedding, input_tokens,
                           num_return_sequences, max_generate_length, typical_sampling, typical_mass, conds_latent, hf_generate_kwargs):
        """Argument handling of `inference_speech` up to the `generate` call (model_v2.py:716-803)."""
        if input_tokens is not None or num_return_sequences != 1:
            raise NotImplementedError("input_tokens / num_return_sequences > 1 are not used by the v2.5 pipeline")
        if typical_sampling and not (typical_mass > 0.0 and typical_mass < 1.0):           # model_v2.py:796-797
            raise ValueError(f"`typical_mass` has to be a float > 0 and < 1, but is {typical_mass}")
        if conds_latent is None:
            if emo_vec is None:
                raise NotImplementedError("emo_vec=None needs the Conformer/Perceiver emotion encoder (PyTorch side): "
                                          "compute it with merge_emovec / get_emovec and pass emo_vec=")
            if self.spk_cond_mode == "campplus":
                if campplus_embedding is None:
                    raise ValueError("campplus mode requires campplus_embedding or wav")
                conds_latent, spk_lat = self.conds_latent(campplus_embedding, emo_vec)
            else:                                                   # IndexTTS-2 (model_v2.py:761,767-773)
                if speech_condition.ndim == 2:
                    speech_condition = speech_condition.unsqueeze(0)
                if cond_lengths is None:       # model_v2.py:761 passes the feature width; = "all frames valid" (clamped to the frame count)
                    cond_lengths = torch.tensor([min(int(speech_condition.shape[-1]), int(speech_condition.shape[1]))],
                                                device=speech_condition.device)
                spk_lat = self.get_conditioning(speech_condition.transpose(1, 2), cond_lengths)
                conds_latent = self.conds_latent_v2(spk_lat, emo_vec)
                langs = None
        else:
            spk_lat = conds_latent[:, :1] if self.spk_cond_mode == "campplus" else conds_latent[:, : self.cond_num]
        input_ids, inputs_embeds, attention_mask = self.prepare_gpt_inputs(conds_latent, text_inputs, langs)
        max_new = (self.max_mel_tokens - 1) if max_generate_length is None else int(max_generate_length)
        hf = dict(hf_generate_kwargs)
        _reject_logits_processor(hf)
        hf["typical_mass"] = float(typical_mass) if typical_sampling else 0.0
        return inputs_embeds, attention_mask, max_new, hf, spk_lat

    def inference_speech(self, speech_condition, text_inputs, langs=None, emo_speech_condition=None, cond_lengths=None,
                         emo_cond_lengths=None, emo_vec=None, use_speed=False, campplus_embedding=None, wav=None,
                         input_tokens=None, num_return_sequences=1, max_generate_length=None, typical_sampling=False,
                         typical_mass=.9, conds_latent=None, uniforms=None, **hf_generate_kwargs):
        """model_v2.py:716-825 (campplus conditioning).  Returns (codes, speech_conditioning_latent)."""
        inputs_embeds, attention_mask, max_new, hf, spk_lat = self._prepare_inference(
            speech_condition, text_inputs, langs, cond_lengths, emo_vec, campplus_embedding, input_tokens, num_return_sequences,
            max_generate_length, typical_sampling, typical_mass, conds_latent, hf_generate_kwargs)
        codes = self.generate(inputs_embeds, attention_mask, max_new, uniforms=uniforms, **hf)
        return codes, spk_lat

    def inference_speech_stream(self, speech_condition, text_inputs, chunk_size=100, overlap_size=20, langs=None, cond_lengths=None,
                                emo_vec=None, campplus_embedding=None, max_generate_length=None, typical_sampling=False,
                                typical_mass=.9, conds_latent=None, uniforms=None, **hf_generate_kwargs):
        """`inference_speech` whose generate call yields code chunks as they are decoded (see `generate_chunks`): returns the
        (inputs_embeds, attention_mask, max_new_tokens, generate kwargs) a `streaming.StreamingDecoder` feeds back to this engine."""
        inputs_embeds, attention_mask, max_new, hf, _ = self._prepare_inference(
            speech_condition, text_inputs, langs, cond_lengths, emo_vec, campplus_embedding, None, 1, max_generate_length,
            typical_sampling, typical_mass, conds_latent, hf_generate_kwargs)
        hf["uniforms"] = uniforms
        return inputs_embeds, attention_mask, max_new, hf

    def inference_speech_inflight(self, speech_condition, text_inputs, langs=None, cond_lengths=None, emo_vec=None, campplus_embedding=None,
                                  max_generate_length=None, typical_sampling=False, typical_mass=.9, conds_latent=None, slots=8,
                                  chunk_tokens=16, min_free=1, row_max_new: Optional[Sequence[int]] = None, **hf_generate_kwargs):
        """`inference_speech` for MORE utterances than decode slots: `slots` rows decode at a time and, whenever rows have emitted their stop
        token, waiting utterances are prefilled into the freed slots (`DecodeSession.admit`) instead of waiting for the whole batch to drain --
        the in-flight batching of the reference's serving path (backends/trt/serving/triton_server.py:96-305, pipeline.py:459-548) as a
        scheduling loop around the engine's suspended decode loop.  A row's ids do not depend on the batch it runs in or on when it joins
        (greedy: bit for bit the ids of `inference_speech` over all utterances at once; sampling: slot- and row-step-keyed random stream).

        ONE session serves the whole call: every slot keeps its own cache position and its own step (position embedding, token cap), so an
        utterance can join at any step with its full budget (`max_generate_length`, or its `row_max_new` cap) -- nothing of the session is
        bounded by the mel position table, only each row is.  Rows are polled every `chunk_tokens` tokens; an admission (one prefill launch
        train for all the utterances it places) waits until `min_free` slots are free -- or nothing is running.  `row_max_new`: per-utterance
        token caps as in `generate`.  Returns (codes (N, L) padded with the stop token, speech_conditioning_latent); `last_inflight` holds the
        schedule's counters.  num_beams = 1."""
        emb, mask, max_new, hf, spk_lat = self._prepare_inference(
            speech_condition, text_inputs, langs, cond_lengths, emo_vec, campplus_embedding, None, 1, max_generate_length, typical_sampling,
            typical_mass, conds_latent, hf_generate_kwargs)
        if hf.get("num_beams", 1) != 1:
            raise NotImplementedError("inference_speech_inflight: num_beams = 1 only")
        N, slots, chunk = emb.shape[0], max(1, int(slots)), max(1, int(chunk_tokens))
        table = int(self._emb["mel_pos_embedding.emb.weight"].shape[0]) + 1 - (2 if self.kv_cache else 1)      # the engine's bound on a ROW's steps
        if max_new > table:
            raise ValueError(f"max_generate_length = {max_new} exceeds the mel position table ({table} steps)")
        stop = self.stop_mel_token
        if row_max_new is not None and len(row_max_new) != N:
            raise ValueError(f"row_max_new must have one entry per utterance ({N}), got {len(row_max_new)}")
        cap = [max_new if row_max_new is None else max(0, min(max_new, int(v))) for v in (row_max_new if row_max_new is not None else range(N))]
        caps_of = lambda idx: [cap[i] for i in idx]          # always enforced by the engine's sampler: a row at its cap emits the stop token, which
                                                               # is what frees its slot (admission needs the engine to have the row as finished)
        min_free = max(1, int(min_free))
        results: List[Optional[torch.Tensor]] = [None] * N
        stats = dict(sessions=1, admitted=0, admissions=0, steps=0, row_steps=0, truncated=0)
        first, pending = list(range(N))[:slots], list(range(N))[slots:]
        B = len(first)
        owner: List[Optional[int]] = list(first)
        with DecodeSession(self, emb[first], mask[first], max_new, row_max_new=caps_of(first), **hf) as sess:
            while any(o is not None for o in owner):
                before = sess.steps
                # while utterances wait, come back as soon as `min_free` slots can be refilled (the engine looks at its flags every 8 steps)
                sess.run(chunk, return_when_finished=min(min_free, len(pending), B) if pending else 0)
                stats["row_steps"] += (sess.steps - before) * sum(o is not None for o in owner)
                if sess.steps == before:
                    raise _lib.HipEngineError("inference_speech_inflight: the decode session made no progress")
                for b, n_codes in sess.finished_lengths():           # one device reduction + one host synchronisation per poll
                    if owner[b] is None:
                        continue
                    c = sess._codes[b, :min(n_codes, cap[owner[b]])].clone()
                    if c.numel() >= cap[owner[b]]:
                        stats["truncated"] += 1              # ran into its cap before a stop token of its own
                    if c.numel() < max_new:
                        c = torch.cat([c, c.new_full((1,), stop)])
                    results[owner[b]] = c
                    owner[b] = None
                free = [b for b in range(B) if owner[b] is None]
                enough = len(free) >= min(min_free, len(pending)) or len(free) == B
                if free and pending and enough:
                    take, pending = pending[:len(free)], pending[len(free):]
                    free = free[:len(take)]
                    sess.admit(free, emb[take], mask[take], row_max_new=caps_of(take))
                    for b, i in zip(free, take):
                        owner[b] = i
                    stats["admitted"] += len(take)
                    stats["admissions"] += 1
            stats["steps"] = sess.steps
        self.last_inflight = stats
        width = max(int(c.numel()) for c in results)
        codes = torch.full((N, width), stop, dtype=torch.int64, device=self.device)
        for i, c in enumerate(results):
            codes[i, : c.numel()] = c
        return codes, spk_lat

    # ---- teacher-forced latent pass (model_v2.py:596-646) ----------------------------------------------------------
    def forward_latent(self, conds: torch.Tensor, text_inputs: torch.Tensor, text_lengths: torch.Tensor,
                       mel_codes: torch.Tensor, mel_codes_lengths: torch.Tensor) -> torch.Tensor:
        self._check_idle("forward_latent")
        dev = self.device
        text = text_inputs.to(dev).long().clone()
        mel = mel_codes.to(dev).long().clone()
        tl, ml = text_lengths.to(dev), mel_codes_lengths.to(dev)
        text = torch.where(torch.arange(text.shape[1], device=dev)[None] >= tl[:, None],
                           torch.full_like(text, self.stop_text_token), text)
        mel = torch.where(torch.arange(mel.shape[1], device=dev)[None] >= ml[:, None],
                          torch.full_like(mel, self.stop_mel_token), mel)
        text = F.pad(F.pad(text, (0, 1), value=self.stop_text_token), (1, 0), value=self.start_text_token)
        mel = F.pad(F.pad(mel, (0, 1), value=self.stop_mel_token), (1, 0), value=self.start_mel_token)
        te = self._emb["text_embedding.weight"][text] + self._emb["text_pos_embedding.emb.weight"][: text.shape[1]]
        me = self._emb["mel_embedding.weight"][mel] + self._emb["mel_pos_embedding.emb.weight"][: mel.shape[1]]
        x = torch.cat([conds.to(dev, torch.float32), te, me], dim=1).contiguous()
        B, S, D = x.shape
        L = _lib.lib()
        ws = self._workspace(L.itts_gpt_workspace_bytes(self._h, B, S, S))
        out = torch.empty_like(x)
        _lib.check(L.itts_gpt_forward_latent(self._h, _lib.ptr(x), B, S, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                             _lib.stream_ptr(self.device)), "itts_gpt_forward_latent")
        enc = out[:, conds.shape[1]:]
        return enc[:, -mel.shape[1]:][:, :-2]

    def forward(self, speech_conditioning_latent, text_inputs, text_lengths, mel_codes, mel_codes_lengths,
                emo_speech_conditioning_latent=None, cond_mel_lengths=None, emo_cond_mel_lengths=None, emo_vec=None,
                use_speed=None, do_spk_cond=False):
        """`UnifiedVoice.forward` of v2/v2.5 (model_v2.py:596-646; call site infer_v2.py:636-651): the teacher-forced
        pass that returns the mel-position latents.  campplus conditioning: `speech_conditioning_latent` is the style
        vector when `do_spk_cond` (projected here) or the projected (b,1,D) latent otherwise; `emo_vec` must be given
        (the emotion Conformer/Perceiver is outside this path)."""
        if emo_vec is None:
            raise NotImplementedError("emo_vec=None needs the emotion encoder (PyTorch side); pass emo_vec=")
        dev = self.device
        spk = speech_conditioning_latent.to(dev, torch.float32)
        if self.spk_cond_mode != "campplus":                       # IndexTTS-2: latents (b, 32, D) + speed embeddings (:634-637)
            if do_spk_cond:
                spk = self.get_conditioning(spk.transpose(1, 2), cond_mel_lengths).to(dev, torch.float32)
            se = self._emb["speed_emb.weight"]
            us = torch.zeros(spk.shape[0], dtype=torch.long, device=dev) if use_speed is None else torch.as_tensor(use_speed).to(dev).long()
            dur, half = se[torch.zeros_like(us)], se[torch.ones_like(us)]
            conds = torch.cat((spk + emo_vec.to(dev, torch.float32).unsqueeze(1), half.unsqueeze(1), dur.unsqueeze(1)), 1)
            return self.forward_latent(conds, text_inputs, text_lengths, mel_codes, mel_codes_lengths)
        if do_spk_cond:
            spk = _spk_proj(spk, *self._spk_proj_params())
            if spk.ndim != 3:
                spk = spk.unsqueeze(1)
        conds = torch.cat((spk + emo_vec.to(dev, torch.float32).unsqueeze(1),
                           torch.zeros(spk.size(0), 2, spk.size(2), device=dev)), 1)
        return self.forward_latent(conds, text_inputs, text_lengths, mel_codes, mel_codes_lengths)

    __call__ = forward

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _lib.lib().itts_gpt_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class DecodeSession:
    """A decode batch that keeps running while its utterances finish and NEW utterances are admitted into the freed slots (design reference: the
    in-flight batching of the reference's serving path, backends/trt/serving/triton_server.py:96-305, backends/trt/pipeline/pipeline.py:459-548).
    Built on the suspended-loop API of the engine: `run(n)` advances every live row by n tokens (`itts_gpt_generate_chunk`), `finished()` reports
    the slots whose row has emitted its stop token, `admit(slots, ...)` prefills new prompts into those slots (`itts_gpt_admit_rows`).  Every slot
    keeps its OWN cache position and its OWN step: an admitted utterance's keys sit at positions 0 .. of its cache row, its codes start at column
    0 of its code row, its position embeddings, random stream and token cap count from its first token -- so it generates, bit for bit, the ids
    it generates decoded alone, whatever step it joins at, and the session runs for as long as utterances keep arriving (only a ROW is bounded,
    by `max_new_tokens`).  Greedy / sampling, num_beams = 1.

    inputs_embeds (B, s, D) / attention_mask (B, s + 1): what `UnifiedVoice.inference_speech_stream` returns for the first batch."""

    def __init__(self, model: "UnifiedVoice", inputs_embeds: torch.Tensor, attention_mask: torch.Tensor, max_new_tokens: int, do_sample=False,
                 top_p=1.0, top_k=50, temperature=1.0, repetition_penalty=1.0, length_penalty=1.0, seed: Optional[int] = None,
                 typical_mass: float = 0.0, row_max_new: Optional[Sequence[int]] = None, uniforms=None, **unused):
        """row_max_new: per-utterance caps on generated tokens (`generate`'s engine extension): the utterance in a slot emits the stop token from
        its own token index row_max_new[b] on; `admit(..., row_max_new=)` passes the caps of the utterances it places."""
        model._check_idle("DecodeSession")
        if unused.get("num_beams", 1) != 1:
            raise NotImplementedError("DecodeSession: num_beams = 1 only")
        if uniforms is not None:        # a uniform stream is laid out per (step, row) of ONE batch; slots here change utterances
            raise NotImplementedError("DecodeSession: `uniforms` is not supported (rows are re-occupied); use `seed`")
        altering = sorted(k for k in unused if k in _UNSUPPORTED_GENERATE_KWARGS and unused[k] is not None)
        if altering:                    # as `generate`: never drop kwargs that change the ids silently
            raise NotImplementedError(f"DecodeSession: {altering} would change the generated ids and the device loop does not implement them")
        self.m, self.dev = model, model.device
        B, s, D = inputs_embeds.shape
        self.B, self.D, self.max_new = B, D, int(max_new_tokens)
        self._start = (model._emb["mel_embedding.weight"][model.start_mel_token] + model._emb["mel_pos_embedding.emb.weight"][0])
        self._x = torch.cat([inputs_embeds.to(self.dev, torch.float32), self._start.expand(B, 1, D)], dim=1).contiguous()
        self.S = s + 1
        self._pad = (attention_mask[:, :self.S] == 0).sum(dim=1).to(torch.int32).to(self.dev).contiguous()
        gp = _lib.GenParams()
        gp.do_sample, gp.num_beams, gp.top_k = int(bool(do_sample)), 1, int(top_k or 0)
        gp.min_tokens_to_keep, gp.max_new_tokens = 1, self.max_new
        gp.pos_offset = 2 if model.kv_cache else 1
        gp.top_p, gp.temperature = float(top_p), float(temperature)
        gp.repetition_penalty = float(repetition_penalty if repetition_penalty is not None else 1.0)
        gp.length_penalty, gp.seed = float(length_penalty), model._seed(seed, do_sample, None)
        gp.typical_mass = float(typical_mass)
        self._gp = gp
        L = _lib.lib()
        self._ws = model._workspace(L.itts_gpt_workspace_bytes(model._h, B, self.S, self.S + self.max_new))
        self._codes = model._persistent("codes", (B, self.max_new), torch.int64)
        self._pen = (C.c_int32 * 2)(1, model.start_mel_token)
        self.steps = 0                               # steps of the session (= tokens generated by the rows of the first batch while they run)
        self.step0 = [0] * B                         # the session step at which the utterance currently in each slot produced its first token
        self._first = True
        self._adm_ws = None
        self._lim = None
        if row_max_new is not None:
            if len(row_max_new) != B:
                raise ValueError(f"row_max_new must have one entry per row ({B}), got {len(row_max_new)}")
            self._lim = model._persistent("row_limits", (B,), torch.int32)     # persistent: its address is part of the decode graph's key
            self._lim.copy_(torch.as_tensor([int(v) for v in row_max_new], dtype=torch.int32))
            _lib.check(L.itts_gpt_set_row_limits(model._h, _lib.ptr(self._lim), B), "itts_gpt_set_row_limits")
        model._stream_open = True                    # the workspace holds this session's state until close()

    def run(self, n_tokens: int, return_when_finished: int = 0) -> int:
        """advance the batch by up to n_tokens steps; returns the session's step count (stops early when every row has finished -- or, with
        return_when_finished = k > 0, at the engine's next flag check (every 8 steps) once k slots hold a finished utterance, counting the ones
        that were finished before the call: the caller refills slots without polling in short chunks)"""
        L = _lib.lib()
        limit = self.steps + int(n_tokens) if not self._first else min(self.max_new, int(n_tokens))
        _lib.check(L.itts_gpt_set_chunk_return(self.m._h, max(0, int(return_when_finished))), "itts_gpt_set_chunk_return")
        n = C.c_int32(0)
        _lib.check(L.itts_gpt_generate_chunk(self.m._h, _lib.ptr(self._x) if self._first else None, _lib.ptr(self._pad), self.B, self.S,
                                             C.byref(self._gp), self._pen, 2, None, _lib.ptr(self._codes), limit, C.byref(n), _lib.ptr(self._ws),
                                             self._ws.numel(), int(self.m.use_graph), _lib.stream_ptr(self.dev)), "itts_gpt_generate_chunk")
        self._first = False
        self.steps = int(n.value)
        return self.steps

    def _own_steps(self, slot: int) -> int:
        """tokens the utterance in `slot` has produced so far (its code columns 0 .. that - 1)"""
        return max(0, min(self.max_new, self.steps - self.step0[slot]))

    def codes(self, slot: int) -> torch.Tensor:
        """the codes of the utterance in `slot` so far (up to, not including, its stop token)"""
        row = self._codes[slot, :self._own_steps(slot)]
        stop = (row == self.m.stop_mel_token).nonzero()
        return row[: int(stop[0])].clone() if stop.numel() else row.clone()

    def finished(self) -> List[int]:
        """slots whose utterance has emitted its stop token or used up its max_new_tokens (one device reduction, one host synchronisation)"""
        if self.steps < 1:
            return []
        own = torch.as_tensor([self.steps - self.step0[b] for b in range(self.B)], device=self.dev)
        live_cols = torch.arange(self.max_new, device=self.dev)[None, :] < own[:, None]
        got = ((self._codes == self.m.stop_mel_token) & live_cols).any(dim=1) | (own > self.max_new)      # (past its last column the engine has the row
                                                                                                          # as stopped: the sampler emits the stop token there)
        return [b for b, v in enumerate(got.tolist()) if v]

    def finished_lengths(self) -> List[tuple]:
        """[(slot, number of codes before its stop token)] of the slots `finished()` reports -- both from one device reduction and one host
        synchronisation (the scheduler's poll: `finished()` + `codes()` per slot would synchronise once per slot)"""
        if self.steps < 1:
            return []
        own = torch.as_tensor([self.steps - self.step0[b] for b in range(self.B)], device=self.dev)
        cols = torch.arange(self.max_new, device=self.dev)[None, :]
        stop = (self._codes == self.m.stop_mel_token) & (cols < own[:, None])
        first = torch.where(stop.any(dim=1), stop.to(torch.int32).argmax(dim=1), own.clamp(max=self.max_new))       # first stop column, else all own columns
        fin = stop.any(dim=1) | (own > self.max_new)
        host = torch.stack([fin.to(torch.int64), first.to(torch.int64)]).tolist()
        return [(b, int(host[1][b])) for b in range(self.B) if host[0][b]]

    def admit(self, slots: Sequence[int], inputs_embeds: torch.Tensor, attention_mask: torch.Tensor,
              row_max_new: Optional[Sequence[int]] = None) -> None:
        """put new utterances into finished slots: inputs_embeds (n, s', D) / attention_mask (n, s' + 1) as for the first batch, s' <= the first
        batch's s (a cache row holds that prompt + max_new_tokens)"""
        if self._first or self.steps < 1:
            raise RuntimeError("DecodeSession.admit: run() the first batch before admitting")
        n, s, D = inputs_embeds.shape
        if (row_max_new is not None) != (self._lim is not None):
            raise ValueError("DecodeSession.admit: row_max_new must be given exactly when the session was opened with per-row caps")
        if row_max_new is not None and len(row_max_new) != n:
            raise ValueError(f"row_max_new must have one entry per admitted row ({n}), got {len(row_max_new)}")
        if len(slots) != n:
            raise ValueError(f"DecodeSession.admit: {len(slots)} slots for {n} utterances")
        if s + 1 > self.S:
            raise ValueError(f"DecodeSession.admit: the prompt ({s + 1} positions) is longer than the session's cache rows hold ({self.S})")
        x = torch.cat([inputs_embeds.to(self.dev, torch.float32), self._start.expand(n, 1, D)], dim=1).contiguous()
        pad = (attention_mask[:, :s + 1] == 0).sum(dim=1).to(torch.int32).to(self.dev).contiguous()
        L = _lib.lib()
        need = L.itts_gpt_admit_workspace_bytes(self.m._h, n, s + 1)
        if self._adm_ws is None or self._adm_ws.numel() < need:
            self._adm_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        sl = (C.c_int32 * n)(*[int(v) for v in slots])
        lim = None if row_max_new is None else (C.c_int32 * n)(*[int(v) for v in row_max_new])     # written to the live limits by the engine,
        _lib.check(L.itts_gpt_admit_rows(self.m._h, _lib.ptr(x), _lib.ptr(pad), sl, n, s + 1, lim, C.byref(self._gp), self._pen, 2, None,   # after its checks
                                         _lib.ptr(self._codes), _lib.ptr(self._ws), self._ws.numel(), _lib.ptr(self._adm_ws), self._adm_ws.numel(),
                                         _lib.stream_ptr(self.dev)), "itts_gpt_admit_rows")
        for v in slots:
            self.step0[int(v)] = self.steps - 1

    def close(self):
        _lib.lib().itts_gpt_set_chunk_return(self.m._h, 0)
        if self._lim is not None:
            _lib.lib().itts_gpt_set_row_limits(self.m._h, None, 0)
            self._lim = None
        self.m._stream_open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class UnifiedVoiceV1(UnifiedVoice):
    """IndexTTS-1 / 1.5 `UnifiedVoice` (indextts/gpt/model.py): the same GPT-2 stack and generate loop; conditioning is a
    (b,32,D) latent from the Conformer+Perceiver encoder (`get_conditioning`, model.py:495-524), which stays on PyTorch:
    set `conditioning_fn` (e.g. the reference module's bound `get_conditioning`) or pass `conds_latent=`.
    No language embedding, `inference_speech` returns the codes only (model.py:660-716)."""
    _ALLOW_ENCODER_CONDITIONING = True
    _HOST_TENSORS = ("mel_embedding.weight", "mel_pos_embedding.emb.weight", "text_embedding.weight",
                     "text_pos_embedding.emb.weight")
    _OPTIONAL_HOST_TENSORS = ()

    def __init__(self, *args, condition_type="conformer_perceiver", conditioning_fn=None, **kw):
        kw.setdefault("spk_cond_mode", "encoder")
        super().__init__(*args, condition_type=condition_type, **kw)
        self.condition_type = condition_type
        self.conditioning_fn = conditioning_fn

    def get_conditioning(self, speech_conditioning_input, cond_mel_lengths=None):
        if self.conditioning_fn is None:
            raise NotImplementedError("v1 conditioning encoder (Conformer + Perceiver) is outside the engine: set "
                                      "UnifiedVoiceV1.conditioning_fn or pass conds_latent=")
        return self.conditioning_fn(speech_conditioning_input, cond_mel_lengths)

    def prepare_gpt_inputs(self, conditional_latents, text_inputs):                      # model.py:597-659
        return super().prepare_gpt_inputs(conditional_latents, text_inputs, None)

    def inference_speech(self, speech_conditioning_mel, text_inputs, cond_mel_lengths=None, input_tokens=None,
                         num_return_sequences=1, max_generate_length=None, typical_sampling=False, typical_mass=.9,
                         conds_latent=None, uniforms=None, **hf_generate_kwargs):
        if input_tokens is not None or num_return_sequences != 1:
            raise NotImplementedError("input_tokens / num_return_sequences > 1 are not used by the v1 pipeline")
        if typical_sampling and not (typical_mass > 0.0 and typical_mass < 1.0):
            raise ValueError(f"`typical_mass` has to be a float > 0 and < 1, but is {typical_mass}")
        if conds_latent is None:
            if speech_conditioning_mel.ndim == 2:
                speech_conditioning_mel = speech_conditioning_mel.unsqueeze(0)
            if cond_mel_lengths is None:
                cond_mel_lengths = torch.tensor([speech_conditioning_mel.shape[-1]], device=speech_conditioning_mel.device)
            conds_latent = self.get_conditioning(speech_conditioning_mel, cond_mel_lengths)
        input_ids, inputs_embeds, attention_mask = self.prepare_gpt_inputs(conds_latent, text_inputs)
        max_new = (self.max_mel_tokens - 1) if max_generate_length is None else int(max_generate_length)
        hf = dict(hf_generate_kwargs)
        _reject_logits_processor(hf)
        return self.generate(inputs_embeds, attention_mask, max_new, uniforms=uniforms,
                             typical_mass=float(typical_mass) if typical_sampling else 0.0, **hf)

    def forward(self, speech_conditioning_latent, text_inputs, text_lengths, mel_codes, wav_lengths, cond_mel_lengths=None,
                types=None, text_first=True, raw_mels=None, return_attentions=False, return_latent=False,
                clip_inputs=False, conds_latent=None):
        """model.py:526-590 with `return_latent=True` (the only use on the inference path, infer.py:449-454,638-643)."""
        if not return_latent or not text_first or raw_mels is not None or return_attentions or clip_inputs:
            raise NotImplementedError("only the return_latent=True, text_first inference form is on the engine path")
        if conds_latent is None:
            conds_latent = self.get_conditioning(speech_conditioning_latent, cond_mel_lengths)
        if types is not None:
            text_inputs = text_inputs * (1 + types).unsqueeze(-1)
        wl = torch.as_tensor(wav_lengths)
        mel_codes_lengths = torch.ceil(wl / self.mel_length_compression).long() + 1              # model.py:557
        b = text_inputs.shape[0]
        conds = conds_latent.expand(b, -1, -1) if conds_latent.shape[0] == 1 and b > 1 else conds_latent
        return self.forward_latent(conds, text_inputs, torch.as_tensor(text_lengths), mel_codes, mel_codes_lengths)

    __call__ = forward


def pack_gemm_weight(w_kn: torch.Tensor, precision: int, transposed: bool = False) -> torch.Tensor:
    w = w_kn.detach().to("cpu", torch.float32).contiguous()
    K, N = (w.shape[1], w.shape[0]) if transposed else (w.shape[0], w.shape[1])
    L = _lib.lib()
    out = torch.empty(L.itts_packed_gemm_bytes(K, N, precision), dtype=torch.uint8)
    _lib.check(L.itts_pack_gemm_weight(_lib.ptr(w), K, N, int(transposed), precision, _lib.ptr(out)), "itts_pack_gemm_weight")
    return out


def gemm(a: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], N: int, precision: int,
         prefill_tiles: bool = False) -> torch.Tensor:
    M, K = a.shape
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    with _lib.on_device(a.device):
        _lib.check(_lib.lib().itts_gemm_forward(_lib.ptr(a.contiguous()), _lib.ptr(w_packed), _lib.ptr(bias), _lib.ptr(out),
                                                M, N, K, precision, int(prefill_tiles), 0, _lib.stream_ptr(a.device)),
                   "itts_gemm_forward")
        return out


def linear_f32(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], cache: Optional[dict] = None, key=None) -> torch.Tensor:
    """`F.linear(x, weight, bias)` on the engine's exact-f32 MFMA GEMM (`itts_gemm_forward`, precision 0) -- the small host-side projections of the
    hot path (the flow-matching decoder's per-step conditioning vectors) run on the same hand-written kernels as everything else instead of a
    vendor BLAS call.  `cache` / `key`: a dict OWNED BY THE CALLER'S MODEL and the parameter's name -- the packed weight is kept there, so it lives
    and dies with the model that owns the parameter (no process-wide table keyed by addresses); without a cache the weight is packed per call.
    K not a multiple of 16 (no shape of the shipped models): `F.linear`."""
    N, K = weight.shape
    if K % 16:
        return F.linear(x, weight.to(x.device), None if bias is None else bias.to(x.device))
    packed = cache.get(key) if cache is not None and key is not None else None
    if packed is None or packed.device != x.device:
        packed = pack_gemm_weight(weight, 0, transposed=True).to(x.device)
        if cache is not None and key is not None:
            cache[key] = packed
    lead = x.shape[:-1]
    a = x.reshape(-1, K).to(torch.float32).contiguous()
    out = gemm(a, packed, None if bias is None else bias.to(x.device, torch.float32).contiguous(), N, 0)
    return out.reshape(*lead, N)


def _spk_proj(x: torch.Tensor, w64: torch.Tensor, b64: torch.Tensor) -> torch.Tensor:
    """spk_emb_proj (model_v2.py:754): a (B, 192) x (192, D) projection, once per batch, outside every loop.  Evaluated in float64 on the host and
    rounded once: the result does not depend on any library's summation order.  That matters: the reference-minted `typical_greedy` fixture holds a
    token whose margin is below f32 summation noise of this projection -- the engine's f32 MFMA GEMM (another order than the reference's CPU sgemm)
    flips it, the correctly rounded value keeps it (profiles/r05d/status.txt).  No vendor BLAS call either way.  w64 / b64: the float64 host copies
    `UnifiedVoice._spk_proj_params()` keeps per model (made once, not per call); only the (B, 192) input crosses to the host."""
    y = x.detach().to("cpu", torch.float64) @ w64.t() + b64
    return y.to(torch.float32).to(x.device)


def gemm_ln(x, g, b, w_packed, bias, N: int, partial=None, bias_prev=None, eps=1e-5):
    """The LayerNorm-fused decode GEMM as a unit op (`itts_gemm_ln_forward`): 1-4 rows, bf16-packed weights.  Returns (out (M, N) f32, x' (M, K))."""
    M, K = x.shape
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    x_out = torch.empty_like(x) if partial is not None else None
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().itts_gemm_ln_forward(_lib.ptr(x.contiguous()), _lib.ptr(partial), _lib.ptr(bias_prev), _lib.ptr(g), _lib.ptr(b),
                                                   float(eps), _lib.ptr(w_packed), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(x_out), M, N, K,
                                                   _lib.stream_ptr(x.device)), "itts_gemm_ln_forward")
    return out, x_out


def layernorm(x, g, b, g2=None, b2=None, eps=1e-5):
    rows, D = x.shape
    out = torch.empty_like(x)
    with _lib.on_device(x.device):
        _lib.check(_lib.lib().itts_layernorm_forward(_lib.ptr(x.contiguous()), _lib.ptr(g), _lib.ptr(b), _lib.ptr(g2),
                                                     _lib.ptr(b2), _lib.ptr(out), rows, D, float(eps), _lib.stream_ptr(x.device)),
                   "itts_layernorm_forward")
        return out
