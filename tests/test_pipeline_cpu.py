"""CPU control-flow tests of the IndexTTS2 boundary class with fake engines (no GPU): segment batching, stop-token trim,
silence insertion, return formats, WAV writer."""
import os
import wave

import numpy as np
import torch

from indextts_amd.infer_v2_5 import IndexTTS2, save_pcm_wav
from tests.pipeline_stubs import StubFrontend


class FakeGPT:
    n_text_pos = 42

    def __init__(self):
        self.calls = []

    def inference_speech(self, cond, text, langs, **kw):
        self.calls.append((text.clone(), kw))
        B = text.shape[0]
        codes = torch.full((B, 9), 8193, dtype=torch.long)
        for b in range(B):
            n = 3 + int((text[b] != 1).sum()) % 5          # 3..7 codes then stop tokens
            codes[b, :n] = torch.arange(10 * b, 10 * b + n)
        return codes, None


class FakeVoc:
    total_up = 256

    def __call__(self, mel, lens=None):
        B, _, T = mel.shape
        w = torch.zeros(B, 1, T * 256)
        for b in range(B):
            w[b, :, : int(lens[b]) * 256] = 0.5 + 0.01 * b
        return w


def make(tmp=None):
    fe = StubFrontend(64)
    tts = IndexTTS2(cfg={"gpt": {"stop_mel_token": 8193}}, device="cpu", frontend=fe, gpt=FakeGPT(), bigvgan=FakeVoc())
    return tts, fe


def test_infer_batches_segments_and_returns_gradio_tuple():
    tts, fe = make()
    sr, wav = tts.infer("spk.wav", "hello there. how are you. fine", None, "en", num_beams=1, top_k=1)
    assert sr == 22050 and wav.dtype == np.int16 and wav.ndim == 2 and wav.shape[1] == 1
    assert len(tts.gpt.calls) == 1 and tts.gpt.calls[0][0].shape[0] == 3            # one GPT batch for 3 segments
    kw = tts.gpt.calls[0][1]
    assert kw["do_sample"] is True and kw["num_beams"] == 1 and kw["top_k"] == 1 and kw["repetition_penalty"] == 10.0
    assert kw["max_generate_length"] == 1500 and kw["length_penalty"] == 0.0 and kw["temperature"] == 0.8
    sil = int(22050 * 0.2)
    text = tts.gpt.calls[0][0]
    n_codes = [3 + int((text[b] != 1).sum()) % 5 for b in range(3)]
    expect = sum(2 * n * 256 for n in n_codes) + 2 * sil
    assert wav.shape[0] == expect
    assert abs(int(wav[10, 0]) - int(0.5 * 32767)) <= 1                              # clamp(32767*wav) scaling
    # speaker conditioning cached by path (:620)
    tts.infer("spk.wav", "again", None, "en")
    assert [c for c in fe.calls if c[0] == "speaker"] == [("speaker", "spk.wav")]
    tts.infer("other.wav", "again", None, "en")
    assert len([c for c in fe.calls if c[0] == "speaker"]) == 2


def test_infer_writes_wav_and_stream_return(tmp_path):
    tts, _ = make()
    out = str(tmp_path / "sub" / "o.wav")
    assert tts.infer("spk.wav", "one. two", out, "en") == out
    with wave.open(out) as f:
        assert f.getframerate() == 22050 and f.getsampwidth() == 2 and f.getnchannels() == 1 and f.getnframes() > 0
    chunks = list(tts.infer("spk.wav", "one. two", None, "en", stream_return=True))
    assert len(chunks) == 4 and chunks[1].shape[1] == int(22050 * 0.2)               # wav, silence, wav, silence
    assert tts.infer("spk.wav", " . ", None, "en") is None                             # empty text -> None (:566-567)


def test_infer_batch_groups_utterances():
    tts, _ = make()
    res = tts.infer_batch("spk.wav", ["a. b", "c", "d. e. f"], "en", num_beams=1)
    assert len(res) == 3 and all(r[0] == 22050 for r in res)
    assert tts.gpt.calls[-1][0].shape[0] == 6


def test_trim_codes_and_emo_vector_path():
    tts, _ = make()
    codes = torch.tensor([[5, 6, 8193, 8193], [7, 8, 9, 10], [8193, 1, 2, 3]])
    c, lens = tts.trim_codes(codes)
    assert lens.tolist() == [2, 4, 0] and c.shape == (3, 4)
    tts.infer("spk.wav", "x", None, "en", emo_vector=[0.5, 0, 0, 0, 0, 0, 0, 0], emo_alpha=0.5)


def test_save_pcm_wav_roundtrip(tmp_path):
    """tests/test_audio_save.py of the reference pins the 16-bit PCM writer semantics."""
    wav = torch.tensor([[0.0, 32767.0, -32767.0, 16383.5, 40000.0]])
    p = str(tmp_path / "x.wav")
    save_pcm_wav(p, wav, 22050)
    with wave.open(p) as f:
        data = np.frombuffer(f.readframes(5), dtype=np.int16)
    assert data.tolist()[:3] == [0, 32767, -32767] and data[4] == 32767 and abs(int(data[3]) - 16384) <= 1


class FakeGPTV2(FakeGPT):
    """IndexTTS-2 call sequence: get_conditioning, conds_latent_v2, inference_speech(conds_latent=), then the latent pass"""
    spk_cond_mode = "conformer"

    def get_conditioning(self, x, lengths=None):
        return torch.ones(1, 32, 8)

    def conds_latent_v2(self, lat, emo):
        return torch.cat([lat + emo[:, None, :8], torch.zeros(lat.shape[0], 2, 8)], 1)

    def inference_speech(self, cond, text, langs=None, **kw):
        assert kw["conds_latent"].shape[1:] == (34, 8) and kw["conds_latent"].shape[0] == text.shape[0]
        codes, _ = super().inference_speech(cond, text, langs, **kw)
        return codes, kw["conds_latent"][:, :32]

    def __call__(self, lat, text, text_lens, codes, code_lens, emo_cond, cond_mel_lengths=None, emo_cond_mel_lengths=None, emo_vec=None,
                 use_speed=None, do_spk_cond=False):
        assert lat.shape[0] == text.shape[0] == codes.shape[0] == emo_vec.shape[0] and use_speed.shape[0] == text.shape[0]
        self.latent_calls = getattr(self, "latent_calls", 0) + 1
        return torch.ones(codes.shape[0], codes.shape[1], 8)


def test_v2_pipeline_control_flow():
    """indextts_amd.infer_v2.IndexTTS2 (IndexTTS-2): decode -> trim -> latent pass -> content features -> vocoder, one batch for
    all segments; with no engine s2mel stages injected it goes through the frontend's codes_latent_to_mel hook."""
    from indextts_amd.infer_v2 import IndexTTS2 as V2
    fe = StubFrontend(64)
    g = FakeGPTV2()
    tts = V2(cfg={"gpt": {"stop_mel_token": 8193}, "version": 2.0}, device="cpu", frontend=fe, gpt=g, bigvgan=FakeVoc())
    assert tts.model_version == 2.0 and tts.use_fp16 is False
    sr, wav = tts.infer("spk.wav", "hello there. how are you. fine", None, "en", num_beams=1)
    assert sr == 22050 and wav.dtype == np.int16 and wav.shape[0] > 0
    assert 1 <= g.latent_calls <= 3 and g.calls[0][0].shape[0] == 3      # one unpadded latent pass per (text length, code length) group
    assert set(tts.last_timing) == {"gpt", "gpt_forward", "s2mel", "bigvgan"}
    res = tts.infer_batch("spk.wav", ["a. b", "c"], "en", num_beams=1)
    assert len(res) == 2 and all(r[0] == 22050 for r in res)


def test_engine_stages_are_built_from_the_frontends_state_dicts(monkeypatch):
    """`IndexTTS2.__init__` builds the codes -> mel engine stages from `frontend.engine_state_dicts()` when the config describes them: a stage
    that was injected is kept; IndexTTS-2 (`USE_GPT_LATENT`) builds `MyModel(use_gpt_latent=True)` and loads `gpt_layer`, and refuses a frontend
    whose dicts lack it.  The engine classes are replaced by recorders (their real constructors need a GPU)."""
    from indextts_amd import codec as codec_mod, s2mel as s2mel_mod
    from indextts_amd.infer_v2 import IndexTTS2 as V2
    made = []

    class RecCodec:
        def __init__(self, **kw):
            self.kw, self.loaded = kw, None
            made.append(("codec", kw))

        def load_state_dict(self, sd):
            self.loaded = sd

    class RecS2:
        def __init__(self, args, use_gpt_latent=False, precision="bf16", device="cuda:0"):
            self.args, self.use_gpt_latent, self.precision, self.net = args, use_gpt_latent, precision, None
            made.append(("s2mel", use_gpt_latent, precision))

        def load_state_dict(self, net):
            self.net = net

    monkeypatch.setattr(codec_mod, "EnhancedCodec", RecCodec)
    monkeypatch.setattr(s2mel_mod, "MyModel", RecS2)

    class FE(StubFrontend):
        def __init__(self, with_gpt_layer):
            super().__init__(64)
            self.with_gpt_layer = with_gpt_layer

        def engine_state_dicts(self):
            d = dict(semantic_codec={"c": 1}, cfm={"f": 2}, length_regulator={"r": 3})
            if self.with_gpt_layer:
                d["gpt_layer"] = {"g": 4}
            return d

    cfg = {"gpt": {"stop_mel_token": 8193}, "s2mel": {"x": 1}, "semantic_codec": {"hidden_size": 64}}
    tts = IndexTTS2(cfg=cfg, device="cpu", frontend=FE(False), gpt=FakeGPT(), bigvgan=FakeVoc())
    assert made == [("codec", {"hidden_size": 64, "device": "cpu"}), ("s2mel", False, "fp32")]
    assert tts.semantic_codec.loaded == {"c": 1} and tts.s2mel.net == {"cfm": {"f": 2}, "length_regulator": {"r": 3}}
    made.clear()
    mine = object()
    tts = V2(cfg=dict(cfg, version=2.0), device="cpu", frontend=FE(True), gpt=FakeGPTV2(), bigvgan=FakeVoc(), semantic_codec=mine, use_fp16=True)
    assert tts.semantic_codec is mine and made == [("s2mel", True, "bf16")]                     # the injected codec is kept
    assert tts.s2mel.net == {"cfm": {"f": 2}, "length_regulator": {"r": 3}, "gpt_layer": {"g": 4}}
    import pytest
    with pytest.raises(RuntimeError, match="gpt_layer"):
        V2(cfg=dict(cfg, version=2.0), device="cpu", frontend=FE(False), gpt=FakeGPTV2(), bigvgan=FakeVoc())
    made.clear()
    IndexTTS2(cfg=cfg, device="cpu", frontend=FE(False), gpt=FakeGPT(), bigvgan=FakeVoc(), semantic_codec=mine, s2mel=mine)
    assert made == []                                                                            # both injected: nothing is built


def test_inflight_slots_route_the_gpt_stage_through_the_inflight_call():
    """`infer_batch(..., num_beams=1, inflight_slots=S)` with more segments than slots -> `inference_speech_inflight(slots=S, ...)`; fewer segments,
    several beams or no `inflight_slots` -> the plain batch call.  The scheduling kwargs never reach the engine's generate kwargs."""
    class GPT(FakeGPT):
        def __init__(self):
            super().__init__()
            self.inflight = []

        def inference_speech_inflight(self, cond, text, langs, slots=None, chunk_tokens=16, min_free=1, **kw):
            self.inflight.append((text.shape[0], slots, chunk_tokens, min_free, kw))
            return FakeGPT.inference_speech(self, cond, text, langs, **kw)

    fe = StubFrontend(64)
    tts = IndexTTS2(cfg={"gpt": {"stop_mel_token": 8193}}, device="cpu", frontend=fe, gpt=GPT(), bigvgan=FakeVoc())
    texts = ["one.", "two two.", "three three three.", "four.", "five five."]
    out = tts.infer_batch("spk.wav", texts, "en", num_beams=1, inflight_slots=2, chunk_tokens=8, min_free=2)
    assert len(out) == 5 and len(tts.gpt.inflight) == 1
    n, slots, chunk, min_free, kw = tts.gpt.inflight[0]
    assert (n, slots, chunk, min_free) == (5, 2, 8, 2) and kw["num_beams"] == 1 and "inflight_slots" not in kw and "chunk_tokens" not in kw
    plain = len(tts.gpt.calls)
    tts.infer_batch("spk.wav", texts, "en", num_beams=1, inflight_slots=8)           # everything fits the slots: one ordinary batch
    tts.infer_batch("spk.wav", texts, "en", num_beams=3, inflight_slots=2)           # the reference's default beam search: ordinary batch
    tts.infer_batch("spk.wav", texts, "en", num_beams=1)
    assert len(tts.gpt.inflight) == 1 and len(tts.gpt.calls) == plain + 3
    for _, kw in tts.gpt.calls[plain:]:
        assert "inflight_slots" not in kw
