"""`ReferenceFrontend` wiring (indextts_amd/infer_v2_5.py) without the reference package: a fake `indextts` package with the
attributes the frontend touches is put into sys.modules, so the call sequences (infer_v2_5.py:620-727) run end to end on CPU;
and without any `indextts` package the constructor fails loudly with ImportError."""
import sys
import types

import pytest
import torch

from indextts_amd.infer_v2_5 import ReferenceFrontend


class _Tok:
    def encode(self, text, allowed_special="all"):
        return [2 + (ord(c) % 50) for c in text if c != " "]


class _TextProc:
    import re as _re
    clean_pattern = _re.compile(r"[“”]")
    char_rep_map = {"“": '"', "”": '"'}

    def normalize(self, t):
        return t.replace("2", "two")


class _Reg(torch.nn.Module):
    def forward(self, x, ylens=None, n_quantizers=None, f0=None):
        return torch.ones(1, int(ylens[0]), 512), ylens, None, None, None


class _CFM(torch.nn.Module):
    def inference(self, mu, x_lens, prompt, style, f0, steps, inference_cfg_rate=0.7):
        return torch.zeros(1, 80, mu.shape[1])


class _S2Mel:
    def __init__(self):
        self.models = {"length_regulator": _Reg(), "cfm": _CFM()}


class _Codec(torch.nn.Module):
    def decode(self, codes):
        return torch.zeros(1, 2 * codes.shape[1], 1024)


class _RefGPT:
    gpt = "stack"
    inference_model = "stack"

    def merge_emovec(self, a, b, la, lb, alpha=1.0):
        return torch.full((1, 1280), float(alpha))


class FakeRefIndexTTS2:
    constructed = []

    def __init__(self, cfg_path="checkpoints/config.yaml", model_dir="checkpoints", use_bf16=False, device=None, use_cuda_kernel=None,
                 **kw):
        FakeRefIndexTTS2.constructed.append((cfg_path, model_dir, use_bf16, device))
        self.tokenizer, self.text_process, self.ja_text_process = _Tok(), _TextProc(), None
        self.gpt, self.bigvgan, self.s2mel, self.semantic_codec = _RefGPT(), object(), _S2Mel(), _Codec()
        self.campplus_model = lambda feat: torch.ones(1, 192)
        self.mel_fn = lambda a: torch.zeros(1, 80, a.shape[-1] // 256)
        self.extract_features = lambda a, sampling_rate=16000, return_tensors="pt": {"input_features": torch.zeros(1, 7, 160),
                                                                                      "attention_mask": torch.ones(1, 7)}
        self.emo_num = [2, 3]
        self.emo_matrix = [torch.ones(2, 1280), 2 * torch.ones(3, 1280)]
        self.spk_matrix = [torch.ones(2, 192), torch.ones(3, 192)]

    def get_emb(self, feats, mask):
        return torch.zeros(1, feats.shape[1], 1024)

    def _load_and_cut_audio(self, path, seconds, verbose=False, sr=None):
        return torch.zeros(1, 16000), (sr or 24000)

    def split_text_by_tokens(self, text, max_tokens, lang_prefix=""):
        return [s.strip() for s in text.split(".") if s.strip()]


@pytest.fixture
def fake_reference(monkeypatch):
    mods = {}
    for name in ("indextts", "indextts.infer_v2_5", "indextts.utils", "indextts.utils.tokenizer", "torchaudio", "torchaudio.transforms",
                 "torchaudio.compliance", "torchaudio.compliance.kaldi"):
        m = types.ModuleType(name)
        m.__path__ = []
        mods[name] = m
    r = mods["indextts.infer_v2_5"]
    r.IndexTTS2 = FakeRefIndexTTS2
    r.find_most_similar_cosine = lambda q, m: 0
    r.nemo_text_normalize = lambda t, lang: t
    r.apply_pronunciation_annotations = lambda t: t
    mods["indextts.utils.tokenizer"].lang_to_token = lambda lang: {"en": 3, "zh": 1}.get(lang.lower(), 0)
    mods["torchaudio.transforms"].Resample = lambda a, b: (lambda x: x)
    mods["torchaudio.compliance.kaldi"].fbank = lambda a, num_mel_bins=80, dither=0, sample_frequency=16000: torch.zeros(50, 80)
    mods["torchaudio"].transforms = mods["torchaudio.transforms"]
    mods["torchaudio"].compliance = mods["torchaudio.compliance"]
    mods["torchaudio.compliance"].kaldi = mods["torchaudio.compliance.kaldi"]
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)
    FakeRefIndexTTS2.constructed.clear()
    return mods


def test_reference_frontend_builds_and_runs_with_an_importable_reference(fake_reference):
    fe = ReferenceFrontend({"gpt": {}}, "ckpt_dir", "cpu", cfg_path="ckpt_dir/config.yaml")
    assert FakeRefIndexTTS2.constructed == [("ckpt_dir/config.yaml", "ckpt_dir", False, "cpu")]
    assert fe.ref.bigvgan is None and fe.ref.gpt.gpt is None and fe.ref.gpt.inference_model is None     # engine replaces these
    b = fe.speaker_bundle("spk.wav")
    assert b["style"].shape == (1, 192) and b["spk_cond_emb"].shape == (1, 7, 1024) and b["ref_mel"].shape[1] == 80
    assert b["prompt_condition"].shape == (1, b["ref_mel"].shape[2], 512)
    assert fe.emo_cond("emo.wav").shape == (1, 7, 1024)
    assert float(fe.merge_emovec(b["spk_cond_emb"], b["spk_cond_emb"], 0.5)[0, 0]) == 0.5
    mat, wsum = fe.emo_vector_mix([0.5, 0.25], b["style"], use_random=False)
    assert mat.shape == (1, 1280) and float(wsum) == 0.75 and float(mat[0, 0]) == 0.5 * 1 + 0.25 * 2
    segs = fe.text_segments("Chapter 2. “Quoted” text", "en", 120, True, 602)
    assert len(segs) == 2 and all(s.dtype == torch.int32 and int(s[-1]) == 1 for s in segs)
    assert fe.lang_id("en") == 3
    mel, lens = fe.codes_to_mel(torch.zeros(2, 6, dtype=torch.long), torch.tensor([6, 4]), b, 1.0)
    assert mel.shape[0] == 2 and mel.shape[1] == 80 and lens.tolist() == [int(12 * 1.72), int(8 * 1.72)]
    sds = fe.engine_state_dicts()
    assert set(sds) == {"semantic_codec", "cfm", "length_regulator"}


def test_reference_frontend_fails_loudly_without_the_reference(monkeypatch):
    for k in [k for k in sys.modules if k == "indextts" or k.startswith("indextts.")]:
        monkeypatch.delitem(sys.modules, k, raising=False)
    monkeypatch.setitem(sys.modules, "indextts", None)             # import indextts -> ImportError
    with pytest.raises(ImportError, match="reference `indextts` package"):
        ReferenceFrontend({"gpt": {}}, "ckpt_dir", "cpu")
