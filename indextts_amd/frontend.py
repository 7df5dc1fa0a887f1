"""`EngineFrontend`: the prompt side of `IndexTTS2` without the reference package.

`indextts.infer_v2_5.IndexTTS2.__init__` (:117-266) builds the prompt-side modules from the checkpoint directory; `ReferenceFrontend`
(indextts_amd/infer_v2_5.py) gets them by constructing that reference object.  This frontend reads the same files itself and puts every
network and every DSP step on the HIP engine:

    hf_cache/w2v-bert-2.0/{config.json, model.safetensors | pytorch_model.bin}  -> indextts_amd.w2vbert.Wav2Vec2BertModel    (:170-176)
    <cfg.w2v_stat> {"mean", "var"}                                               -> semantic_mean, semantic_std               (:177-179)
    hf_cache/campplus_cn_common.bin                                              -> indextts_amd.campplus.CAMPPlus            (:213-221)
    <cfg.s2mel_checkpoint> {"net": {"cfm", "length_regulator", ...}}            -> the prompt's length regulator; the state dicts the
    codec.pth {"model": ...}                                                        pipeline's engine codec / s2mel stages load    (:183-200)
    <cfg.emo_matrix>, <cfg.spk_matrix>, cfg.emo_num                              -> emotion-vector mixing                      (:244-254, 669-680)
    prompt audio (.wav)                                                          -> scipy.io.wavfile + indextts_amd.audio      (:626-648)

What it does NOT restate is the text front end (`TextNormalizer`, the BPE tokenizer, `split_text_by_tokens`: indextts/utils/front.py) and
the QwenEmotion LLM: pass `text_frontend=` (an object with `segments(text, lang, max_text_tokens_per_segment, text_normalization, capacity)
-> list of int32 token tensors` and `lang_id(lang) -> int`), e.g. a thin wrapper over the reference's own tokenizer; without it the text
calls raise.  Audio decoding: the reference calls `librosa.load(path)` (any format, resampled to 22.05 kHz by soxr); here WAV files are
read with scipy and resampled by the engine's windowed-sinc resampler -- the one deliberate numerical difference on this side (librosa
and soxr are not available to pin against), inject `audio_loader=` to use another decoder.
"""
import json
import os
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import audio as A
from .infer_v2_5 import W2V_TAP_LAYER, Frontend

PROMPT_SECONDS = 15                      # `_load_and_cut_audio(path, 15, ...)`, infer_v2_5.py:627,687


def load_wav(path: str) -> Tuple[torch.Tensor, int]:
    """WAV file -> (mono float32 waveform (1, L) in [-1, 1], sample rate).  Integer PCM is scaled by its full range, channels are averaged
    (librosa.load's mono=True)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    x = np.asarray(data)
    if x.dtype == np.uint8:
        y = (x.astype(np.float32) - 128.0) / 128.0
    elif np.issubdtype(x.dtype, np.integer):
        y = x.astype(np.float32) / float(2 ** (8 * x.dtype.itemsize - 1))
    else:
        y = x.astype(np.float32)
    if y.ndim == 2:
        y = y.mean(axis=1)
    return torch.from_numpy(np.ascontiguousarray(y))[None], int(sr)


def _torch_load(path):
    try:
        return torch.load(path, map_location="cpu", weights_only=False)
    except TypeError:                    # older torch without the keyword
        return torch.load(path, map_location="cpu")


def _strip_module(sd):
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def _get(cfg, key, default=None):
    try:
        return cfg[key]
    except (KeyError, TypeError, AttributeError):
        return getattr(cfg, key, default)


class EngineFrontend(Frontend):
    def __init__(self, cfg, model_dir: str, device, gpt_engine=None, text_frontend=None,
                 audio_loader: Optional[Callable[[str], Tuple[torch.Tensor, int]]] = None):
        from .campplus import CAMPPlus
        from .codec import InterpolateRegulator
        from .w2vbert import Wav2Vec2BertModel
        self.cfg, self.model_dir, self.device = cfg, model_dir, torch.device(device)
        self.gpt, self.text, self.load_audio = gpt_engine, text_frontend, audio_loader or load_wav
        self.tokenizer = getattr(text_frontend, "tokenizer", None)
        dev = self.device
        # ---- w2v-bert-2.0 + its statistics (:170-179)
        wdir = os.path.join(model_dir, "hf_cache", "w2v-bert-2.0")
        with open(os.path.join(wdir, "config.json")) as f:
            wc = json.load(f)
        st_path = os.path.join(wdir, "model.safetensors")
        if os.path.isfile(st_path):
            from safetensors.torch import load_file
            wsd = load_file(st_path)
        else:
            wsd = _torch_load(os.path.join(wdir, "pytorch_model.bin"))
        keys = ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "feature_projection_input_dim",
                "position_embeddings_type", "left_max_position_embeddings", "right_max_position_embeddings", "conv_depthwise_kernel_size",
                "hidden_act", "layer_norm_eps", "add_adapter")
        self.tap_layer = min(W2V_TAP_LAYER, int(wc["num_hidden_layers"]))
        self.w2v = Wav2Vec2BertModel(**{k: wc[k] for k in keys if k in wc}, device=dev).load_state_dict(wsd, n_layers=self.tap_layer)
        stats = _torch_load(os.path.join(model_dir, _get(cfg, "w2v_stat")))
        self.semantic_mean = stats["mean"].float().to(dev)
        self.semantic_std = torch.sqrt(stats["var"].float()).to(dev)
        self.features = A.SeamlessM4TFeatureExtractor.from_pretrained(wdir, device=dev)
        # ---- CAMPPlus (:213-221)
        self.campplus = CAMPPlus(feat_dim=80, embedding_size=192, device=dev).load_state_dict(
            _torch_load(os.path.join(model_dir, "hf_cache", "campplus_cn_common.bin")))
        # ---- s2mel checkpoint: the prompt's length regulator here, the whole net for the pipeline's engine stages (:190-206)
        s2 = _get(cfg, "s2mel")
        net = _torch_load(os.path.join(model_dir, _get(cfg, "s2mel_checkpoint")))["net"]
        self._net = {k: _strip_module(v) for k, v in net.items()}
        lr = _get(s2, "length_regulator")
        self.regulator = InterpolateRegulator(
            channels=int(_get(lr, "channels")), sampling_ratios=tuple(_get(lr, "sampling_ratios")), is_discrete=bool(_get(lr, "is_discrete", False)),
            in_channels=_get(lr, "in_channels"), vector_quantize=bool(_get(lr, "vector_quantize", False)),
            codebook_size=int(_get(lr, "content_codebook_size", 1024)), f0_condition=bool(_get(lr, "f0_condition", False)), device=dev)
        self.regulator.load_state_dict(self._net["length_regulator"])
        spect = _get(_get(s2, "preprocess_params"), "spect_params")
        fmax = _get(spect, "fmax", "None")
        self.mel_args = dict(n_fft=int(_get(spect, "n_fft")), win_size=int(_get(spect, "win_length")), hop_size=int(_get(spect, "hop_length")),
                             num_mels=int(_get(spect, "n_mels")), sampling_rate=int(_get(_get(s2, "preprocess_params"), "sr")),
                             fmin=_get(spect, "fmin", 0), fmax=None if fmax == "None" else 8000, center=False)          # :256-265
        # ---- emotion / speaker matrices (:244-254)
        self.emo_num = [int(v) for v in _get(cfg, "emo_num", [])]
        self.emo_matrix = self.spk_matrix = None
        if self.emo_num and _get(cfg, "emo_matrix") and _get(cfg, "spk_matrix"):
            self.emo_matrix = torch.split(_torch_load(os.path.join(model_dir, _get(cfg, "emo_matrix"))).float().to(dev), self.emo_num)
            self.spk_matrix = torch.split(_torch_load(os.path.join(model_dir, _get(cfg, "spk_matrix"))).float().to(dev), self.emo_num)

    # ---- what IndexTTS2.__init__ loads into the engine's codec / s2mel stages --------------------------------------------------
    def engine_state_dicts(self):
        codec_sd = _torch_load(os.path.join(self.model_dir, "codec.pth"))["model"]                                     # EnhancedCodec.load_checkpoint
        return dict(semantic_codec=codec_sd, cfm=self._net["cfm"], length_regulator=self._net["length_regulator"])

    # ---- prompt audio -> speaker bundle (:626-667) -----------------------------------------------------------------------------------
    def _load_and_cut(self, path, sr_target: Optional[int]):
        """`_load_and_cut_audio` (:397-409): mono audio at `sr_target` (default 22050, librosa.load's default), at most 15 s"""
        audio, sr = self.load_audio(path)
        want = int(sr_target or 22050)
        audio = A.Resample(sr, want, device=self.device)(audio.to(self.device))
        return audio[:, : PROMPT_SECONDS * want], want

    @torch.no_grad()
    def _w2v(self, audio_16k):
        inputs = self.features(audio_16k, sampling_rate=16000, return_tensors="pt")
        return self.w2v.get_emb(inputs["input_features"], inputs["attention_mask"], self.semantic_mean, self.semantic_std, layer=self.tap_layer)

    @torch.no_grad()
    def speaker_bundle(self, spk_audio_prompt) -> Dict[str, torch.Tensor]:
        audio, sr = self._load_and_cut(spk_audio_prompt, None)
        audio_22k = A.Resample(sr, 22050, device=self.device)(audio)
        audio_16k = A.Resample(sr, 16000, device=self.device)(audio)
        spk_cond_emb = self._w2v(audio_16k)
        ref_mel = A.mel_spectrogram(audio_22k, **self.mel_args)
        feat = A.subtract_mean(A.fbank(audio_16k, num_mel_bins=80, dither=0, sample_frequency=16000))
        style = self.campplus(feat.unsqueeze(0))
        prompt_condition = self.regulator(spk_cond_emb, ylens=torch.tensor([ref_mel.size(2)]), n_quantizers=3, f0=None)[0]
        return dict(style=style, spk_cond_emb=spk_cond_emb, ref_mel=ref_mel, prompt_condition=prompt_condition)

    @torch.no_grad()
    def emo_cond(self, emo_audio_prompt) -> torch.Tensor:                                  # :682-697
        audio, _ = self._load_and_cut(emo_audio_prompt, 16000)
        return self._w2v(audio)

    def merge_emovec(self, spk_cond_emb, emo_cond_emb, alpha: float) -> torch.Tensor:      # :759-765
        if self.gpt is None or getattr(self.gpt, "cond_encoders", None) is None:
            raise RuntimeError("EngineFrontend.merge_emovec needs the engine GPT with its conditioning encoders loaded (gpt_engine=)")
        return self.gpt.merge_emovec(spk_cond_emb, emo_cond_emb, torch.tensor([min(spk_cond_emb.shape[-1], spk_cond_emb.shape[1])]),
                                     torch.tensor([min(emo_cond_emb.shape[-1], emo_cond_emb.shape[1])]), alpha=alpha)

    def emo_vector_mix(self, emo_vector, style, use_random: bool):                         # :669-680
        import random
        if self.emo_matrix is None:
            raise RuntimeError("EngineFrontend: emo_matrix / spk_matrix / emo_num are not in the checkpoint directory's config")
        w = torch.tensor(emo_vector, dtype=torch.float32, device=self.device)
        if use_random:
            idx = [random.randint(0, n - 1) for n in self.emo_num]
        else:                                                                              # find_most_similar_cosine, :902-908
            q = style.float().to(self.device)
            idx = [int(torch.argmax(torch.nn.functional.cosine_similarity(q, m, dim=1))) for m in self.spk_matrix]
        mat = torch.cat([m[i].unsqueeze(0) for i, m in zip(idx, self.emo_matrix)], 0)
        return torch.sum(w.unsqueeze(1) * mat, 0).unsqueeze(0), torch.sum(w)

    # ---- text (injected) ------------------------------------------------------------------------------------------------------------------
    def text_segments(self, text: str, lang: str, max_text_tokens_per_segment: int, text_normalization: bool, capacity: int) -> List[torch.Tensor]:
        if self.text is None:
            raise RuntimeError("EngineFrontend has no text front end: pass text_frontend= (normaliser + tokenizer + segment splitter, "
                               "indextts/utils/front.py in the reference)")
        return self.text.segments(text, lang, max_text_tokens_per_segment, text_normalization, capacity)

    def lang_id(self, lang: str) -> int:
        if self.text is None:
            raise RuntimeError("EngineFrontend has no text front end: pass text_frontend=")
        return int(self.text.lang_id(lang))


class EngineFrontendV1:
    """Prompt side of the v1 / v1.5 pipeline (`indextts_amd.infer.IndexTTS`, `FrontendV1` protocol) on the engine: the conditioning mel of
    indextts/infer.py:303-323,529-537 -- load, channel mean, `torchaudio.transforms.Resample(sr, 24000)`, truncation, `MelSpectrogramFeatures`
    -- with the WAV decoded by scipy (the reference uses `torchaudio.load`) and both DSP steps in `indextts_amd.audio`.  The text side
    (`tokenizer`) is injected; `conditioning` is the engine GPT's own `get_conditioning` (indextts_amd/cond.py)."""

    def __init__(self, device, gpt_engine=None, tokenizer=None, audio_loader: Optional[Callable[[str], Tuple[torch.Tensor, int]]] = None):
        self.device, self.gpt, self.tokenizer = torch.device(device), gpt_engine, tokenizer
        self.load_audio = audio_loader or load_wav
        self.mel = A.MelSpectrogramFeatures(device=self.device)

    @torch.no_grad()
    def cond_mel(self, audio_prompt, truncate_seconds: Optional[float] = None) -> torch.Tensor:
        audio, sr = self.load_audio(audio_prompt)                       # (1, L) mono: torch.mean(audio, dim=0, keepdim=True)
        audio = A.Resample(sr, 24000, device=self.device)(audio.to(self.device))
        if truncate_seconds:
            audio = audio[:, : int(truncate_seconds * 24000)]
        return self.mel(audio)

    def conditioning(self, cond_mel: torch.Tensor, cond_mel_lengths: torch.Tensor) -> torch.Tensor:
        if self.gpt is None:
            raise RuntimeError("EngineFrontendV1.conditioning needs the engine GPT (gpt_engine=)")
        return self.gpt.get_conditioning(cond_mel, cond_mel_lengths)


class EngineFrontendV2(EngineFrontend):
    """IndexTTS-2 (`indextts/infer_v2.py`) prompt side: as `EngineFrontend`, plus the semantic codec with its encoder half
    (`build_semantic_codec(cfg.semantic_codec)` + `safetensors.torch.load_model(..., aux_paths["semantic_codec"])`, infer_v2.py:138-142) so
    that the prompt's content condition is `length_regulator(quantize(spk_cond_emb)[1])` (:465-479) instead of v2.5's
    `length_regulator(spk_cond_emb)`.  `self.codec` is an `indextts_amd.codec.EnhancedCodec` holding both halves: hand it to the pipeline as
    `IndexTTS2(..., frontend=fe, semantic_codec=fe.codec)` so the weights live on the GPU once."""

    def __init__(self, cfg, model_dir: str, device, gpt_engine=None, text_frontend=None, audio_loader=None, semantic_codec_path: Optional[str] = None):
        super().__init__(cfg, model_dir, device, gpt_engine=gpt_engine, text_frontend=text_frontend, audio_loader=audio_loader)
        from safetensors.torch import load_file
        from .codec import EnhancedCodec
        path = semantic_codec_path or os.path.join(model_dir, "hf_cache", "semantic_codec", "model.safetensors")
        self._codec_sd = load_file(path)
        sc = _get(cfg, "semantic_codec")
        keys = ("codebook_size", "hidden_size", "codebook_dim", "vocos_dim", "vocos_intermediate_dim", "vocos_num_layers")
        self.codec = EnhancedCodec(**{k: int(_get(sc, k)) for k in keys if _get(sc, k) is not None}, device=self.device)
        self.codec.load_state_dict(self._codec_sd)
        if not self.codec._has_encoder:
            raise RuntimeError(f"{path} does not carry the codec's encoder half (down.*, encoder.*, in_project): quantize() is unavailable")

    def engine_state_dicts(self):
        out = dict(semantic_codec=self._codec_sd, cfm=self._net["cfm"], length_regulator=self._net["length_regulator"])
        if "gpt_layer" in self._net:
            out["gpt_layer"] = self._net["gpt_layer"]
        return out

    @torch.no_grad()
    def speaker_bundle(self, spk_audio_prompt) -> Dict[str, torch.Tensor]:
        b = super().speaker_bundle(spk_audio_prompt)
        _, S_ref = self.codec.quantize(b["spk_cond_emb"])
        b["prompt_condition"] = self.regulator(S_ref, ylens=torch.tensor([b["ref_mel"].size(2)]), n_quantizers=3, f0=None)[0]
        return b
