"""Stub frontend for the IndexTTS2 pipeline tests: deterministic tiny stand-ins for the PyTorch-side stages
(the reference tests inject a fake backend the same way: cli_tests/test_cli_v2_batch.py:574-599)."""
import torch

from indextts_amd.infer_v2_5 import Frontend


class StubFrontend(Frontend):
    def __init__(self, model_dim, n_text=200, device="cpu", n_mels=80):
        self.D, self.n_text, self.device, self.n_mels = model_dim, n_text, device, n_mels
        g = torch.Generator().manual_seed(0)
        self.style = torch.randn(1, 192, generator=g)
        self.emo = torch.randn(1, model_dim, generator=g) * 0.1
        self.codebook = torch.randn(8194, n_mels, generator=g)
        self.calls = []

    def speaker_bundle(self, spk_audio_prompt):
        self.calls.append(("speaker", spk_audio_prompt))
        return dict(style=self.style.to(self.device), spk_cond_emb=torch.zeros(1, 4, 1024, device=self.device),
                    ref_mel=torch.zeros(1, self.n_mels, 5, device=self.device),
                    prompt_condition=torch.zeros(1, 5, 512, device=self.device))

    def emo_cond(self, emo_audio_prompt):
        self.calls.append(("emo", emo_audio_prompt))
        return torch.zeros(1, 4, 1024, device=self.device)

    def merge_emovec(self, spk_cond_emb, emo_cond_emb, alpha):
        return (self.emo * float(alpha)).to(self.device)

    def emo_vector_mix(self, emo_vector, style, use_random):
        w = torch.tensor(emo_vector, dtype=torch.float32)
        return torch.ones(1, self.D, device=self.device) * float(w.sum()) * 0.01, float(w.sum())

    def text_segments(self, text, lang, max_text_tokens_per_segment, text_normalization, capacity):
        # one segment per sentence; token = 2 + (ord % (n_text-2)); trailing stop id like F.pad(..., value=1)
        segs = [s for s in text.split(".") if s.strip()]
        out = []
        for s in segs:
            ids = [2 + (ord(ch) % (self.n_text - 2)) for ch in s.strip()][: max_text_tokens_per_segment]
            out.append(torch.tensor(ids + [1], dtype=torch.int32))
        return out

    def lang_id(self, lang):
        return 3

    def codes_to_mel(self, codes, code_lens, bundle, duration_factor):
        # "s2mel": 2 frames per code from a fixed codebook (keeps mel length proportional to code length)
        B = codes.shape[0]
        lens = (code_lens * 2 * duration_factor).long().clamp(min=1)
        T = int(lens.max())
        mel = torch.zeros(B, self.n_mels, T, device=codes.device)
        cb = self.codebook.to(codes.device)
        for b in range(B):
            n = int(code_lens[b])
            if n > 0:
                m = cb[codes[b, :n].clamp(0, 8193)].repeat_interleave(2, dim=0)[: int(lens[b])]
                mel[b, :, : m.shape[0]] = m.t() - 4.0
        return mel, lens.to(torch.int32)


    def codes_latent_to_mel(self, codes, code_lens, latent, bundle):
        """IndexTTS-2 fallback hook (gpt_layer + vq2emb + regulator + CFM on the reference's modules): 1.72 frames per code"""
        B = codes.shape[0]
        lens = (code_lens.float() * 1.72).long().clamp(min=1)
        T = int(lens.max())
        mel = torch.zeros(B, self.n_mels, T, device=codes.device)
        cb = self.codebook.to(codes.device)
        for b in range(B):
            n = int(code_lens[b])
            if n > 0:
                idx = (torch.arange(int(lens[b]), device=codes.device).float() / 1.72).long().clamp(max=n - 1)
                mel[b, :, : int(lens[b])] = (cb[codes[b, :n].clamp(0, 8193)][idx] + 0.01 * latent[b, :n].mean()).t() - 4.0
        return mel, lens.to(torch.int32)


class StubTokenizerV1:
    """Whitespace 'tokenizer' with the TextTokenizer methods the v1 pipeline calls (indextts/utils/front.py)."""

    def __init__(self, n_text=200):
        self.n_text = n_text

    def tokenize(self, text):
        return text.split()

    def split_segments(self, tokens, max_text_tokens_per_segment=120):
        segs, cur = [], []
        for t in tokens:
            cur.append(t.rstrip("."))
            if t.endswith(".") or len(cur) >= max_text_tokens_per_segment:
                segs.append(cur)
                cur = []
        if cur:
            segs.append(cur)
        return segs

    def convert_tokens_to_ids(self, sent):
        return [2 + (sum(map(ord, t)) % (self.n_text - 2)) for t in sent]


class StubFrontendV1:
    """cond mel + conditioning latent stand-ins for the v1 pipeline (indextts/infer.py)."""

    def __init__(self, model_dim, device="cpu", n_text=200):
        g = torch.Generator().manual_seed(3)
        self.mel = torch.randn(1, 100, 37, generator=g)
        self.latent = torch.randn(1, 32, model_dim, generator=g) * 0.3
        self.device = device
        self.tokenizer = StubTokenizerV1(n_text)
        self.calls = []

    def cond_mel(self, audio_prompt, truncate_seconds=None):
        self.calls.append(("cond_mel", audio_prompt, truncate_seconds))
        return self.mel.to(self.device)

    def conditioning(self, cond_mel, cond_mel_lengths):
        return self.latent.to(self.device)
