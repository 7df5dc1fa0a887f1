"""CPU oracle for the autoregressive GPT speech-token decoder  --  TEST INFRASTRUCTURE ONLY.

Plain fp32 torch-on-CPU restatement of the reference decode path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product path (`indextts_amd/`) never does.

PINNING STATUS: PINNED by running the reference's own code in the build container.
  `tools/make_golden_gpt.py` executes, straight from /root/reference, the reference's
  `UnifiedVoice.inference_speech / prepare_gpt_inputs / forward(get_logits)`, its
  `GPT2InferenceModel` (forward + prepare_inputs_for_generation) and its vendored
  `GenerationMixin.generate/_sample/_beam_search` + `BeamSearchScorer`, over the installed
  HF `GPT2Model(eager)` blocks and HF logits processors, and stores the resulting token ids /
  latents as `tests/golden/gpt_*.npz`.  (`indextts/gpt/model_v2.py` as a module is not
  importable under transformers 5.15 -- the pieces are extracted by AST, see tools/ref_shim.py.)
  `tests/test_oracle_gpt.py` checks this restatement against those fixtures bit-for-bit:
  greedy (kv-cache on/off), top-k/top-p sampling, beam search, beam-sample (the reference
  default), ragged left-padded batches, EOS at ragged steps, and the teacher-forced latent pass.
  The only substitution is `torch.multinomial` -> inverse-CDF draws from a stored uniform
  stream (its RNG stream cannot be reproduced on a device); see "Sampling" below.

Reference sources restated (paths relative to the reference repo root):
  UnifiedVoice.inference_speech      indextts/gpt/model_v2.py:716-825
  UnifiedVoice.prepare_gpt_inputs    indextts/gpt/model_v2.py:648-714
  GPT2InferenceModel.forward         indextts/gpt/model_v2.py:121-198  (position rule :145-161)
  prepare_inputs_for_generation      indextts/gpt/model_v2.py:91-119
  UnifiedVoice.forward (latent pass) indextts/gpt/model_v2.py:596-646, get_logits :528-554
  GPT2Block / Attention / MLP        indextts/gpt/transformers_gpt2.py:591-667,129-348,571-585
  attention mask construction        indextts/gpt/transformers_gpt2.py:1053-1067
  GenerationMixin._sample            indextts/gpt/transformers_generation_utils.py:3123-3297
  GenerationMixin._beam_search       indextts/gpt/transformers_generation_utils.py:3325-3609
  _get_logits_processor (order)      indextts/gpt/transformers_generation_utils.py:843-1070
  mask growth per step               indextts/gpt/transformers_generation_utils.py:748-790
  BeamSearchScorer / BeamHypotheses  indextts/gpt/transformers_beam_search.py:215-417,930-1013
  TypicalLogitsWarper                indextts/utils/typical_sampling.py:4-30

Sampling: `torch.multinomial`'s RNG stream cannot be reproduced on a device, so every
sampled mode here draws through an explicit stream of uniforms (`uniforms[step, row, j]`)
with inverse-CDF selection in vocabulary order; the HIP engine consumes the same stream.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class GPTConfig:
    layers: int = 24
    model_dim: int = 1280
    heads: int = 20
    max_text_tokens: int = 600
    max_mel_tokens: int = 1815
    number_text_tokens: int = 12000
    number_mel_codes: int = 8194
    start_mel_token: int = 8192
    stop_mel_token: int = 8193
    start_text_token: int = 0
    stop_text_token: int = 1
    max_conditioning_inputs: int = 1
    n_langs: int = 100 + 1            # len(LANGUAGE_DICT) + 1  (model_v2.py:390)
    types: int = 1
    ln_eps: float = 1e-5              # GPT2Config.layer_norm_epsilon default; nn.LayerNorm default

    @property
    def head_dim(self) -> int:
        return self.model_dim // self.heads

    @property
    def n_mel_pos(self) -> int:       # model_v2.py:399
        return self.max_mel_tokens + 2 + self.max_conditioning_inputs

    @property
    def n_text_pos(self) -> int:      # model_v2.py:400
        return self.max_text_tokens + 2


# ----------------------------------------------------------------------------
# weights (reference state-dict names)
# ----------------------------------------------------------------------------
def synth_weights(cfg: GPTConfig, seed: int = 1234, head_std: float = 0.08) -> Dict[str, torch.Tensor]:
    """Seeded random-init weights: N(0, 0.02) GPT-2 style, LayerNorm gamma ~ 1, small betas.

    `head_std` scales `mel_head` so the top-2 logit margin is >> fp32 round-off
    (SURVEY.md section 8d); residual projections use the GPT-2 1/sqrt(2L) damping.
    """
    g = torch.Generator().manual_seed(seed)
    D, L = cfg.model_dim, cfg.layers
    sd: Dict[str, torch.Tensor] = {}

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    def ln(name):
        sd[name + ".weight"] = 1.0 + rn(D, std=0.05)
        sd[name + ".bias"] = rn(D, std=0.02)

    for i in range(L):
        p = f"gpt.h.{i}."
        ln(p + "ln_1")
        sd[p + "attn.c_attn.weight"] = rn(D, 3 * D, std=0.05)     # HF Conv1D: [in, out]
        sd[p + "attn.c_attn.bias"] = rn(3 * D, std=0.02)
        sd[p + "attn.c_proj.weight"] = rn(D, D, std=0.05 / math.sqrt(2 * L))
        sd[p + "attn.c_proj.bias"] = rn(D, std=0.02)
        ln(p + "ln_2")
        sd[p + "mlp.c_fc.weight"] = rn(D, 4 * D, std=0.05)
        sd[p + "mlp.c_fc.bias"] = rn(4 * D, std=0.02)
        sd[p + "mlp.c_proj.weight"] = rn(4 * D, D, std=0.05 / math.sqrt(2 * L))
        sd[p + "mlp.c_proj.bias"] = rn(D, std=0.02)
    ln("gpt.ln_f")
    ln("final_norm")
    sd["mel_head.weight"] = rn(cfg.number_mel_codes, D, std=head_std)    # nn.Linear: [out, in]
    sd["mel_head.bias"] = rn(cfg.number_mel_codes, std=0.02)
    sd["mel_embedding.weight"] = rn(cfg.number_mel_codes, D, std=0.5)
    sd["mel_pos_embedding.emb.weight"] = rn(cfg.n_mel_pos, D, std=0.3)
    sd["text_embedding.weight"] = rn(cfg.number_text_tokens * cfg.types + 1, D, std=0.5)
    sd["text_pos_embedding.emb.weight"] = rn(cfg.n_text_pos, D, std=0.3)
    sd["lang_embedding.weight"] = rn(cfg.n_langs, D, std=0.3)
    sd["spk_emb_proj.weight"] = rn(D, 192, std=0.05)
    sd["spk_emb_proj.bias"] = rn(D, std=0.02)
    return sd


# ----------------------------------------------------------------------------
# transformer stack (HF GPT2Model with wpe nulled; model_v2.py:259-279)
# ----------------------------------------------------------------------------
def gelu_new(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


KV = List[Tuple[torch.Tensor, torch.Tensor]]

# ----------------------------------------------------------------------------
# numerics modes
#   "f32"  : the reference CPU path (everything fp32) -- the ids-bit-exact contract.
#   "bf16" : the HIP engine's bf16 contract restated on the CPU: GEMM weights, every GEMM input (LayerNorm output,
#            attention output, GELU output, final-norm output) and the K/V cache are rounded to bf16 (round-to-nearest-even,
#            what torch .bfloat16() does); accumulation, biases, the residual stream, the query, LayerNorm statistics, the
#            softmax and the logits stay fp32.  The reference's own bf16 mode (`.bfloat16()` + autocast,
#            indextts/infer_v2_5.py:143-146,758) additionally keeps the residual stream, the query and every GEMM OUTPUT
#            in bf16; tests/golden/gpt_bf16.npz holds its outputs (tools/make_golden_gpt.py bf16) and
#            tests/test_oracle_gpt.py compares the two against the fp32 reference.
# ----------------------------------------------------------------------------
_NUMERICS = "f32"
GEMM_WEIGHTS = ("attn.c_attn.weight", "attn.c_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")


class numerics:
    """Context manager selecting the arithmetic contract of `gpt2_stack` / `lm_head` ("f32" | "bf16")."""

    def __init__(self, mode: str):
        assert mode in ("f32", "bf16"), mode
        self.mode = mode

    def __enter__(self):
        global _NUMERICS
        self.prev, _NUMERICS = _NUMERICS, self.mode
        return self

    def __exit__(self, *a):
        global _NUMERICS
        _NUMERICS = self.prev


def rb(x: torch.Tensor) -> torch.Tensor:
    """round to bf16 and back (identity in f32 mode)"""
    return x.bfloat16().float() if _NUMERICS == "bf16" else x


def bf16_weights(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """State dict with the GEMM weights (the tensors the engine stores in bf16) rounded once; everything else shared."""
    out = dict(sd)
    for k, v in sd.items():
        if k.endswith(GEMM_WEIGHTS) or k == "mel_head.weight":
            out[k] = v.bfloat16().float()
    return out


def gpt2_stack(sd, cfg: GPTConfig, x: torch.Tensor, attention_mask: Optional[torch.Tensor],
               past: Optional[KV]) -> Tuple[torch.Tensor, KV]:
    """x (B,S,D) new positions; attention_mask (B, past+S) of 0/1 or None; returns ln_f(hidden), new KV."""
    B, S, D = x.shape
    H, dh = cfg.heads, cfg.head_dim
    fmin = torch.finfo(x.dtype).min
    add_mask = None
    if attention_mask is not None:
        add_mask = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * fmin        # transformers_gpt2.py:1053-1067
    new_kv: KV = []
    for i in range(cfg.layers):
        p = f"gpt.h.{i}."
        h = rb(F.layer_norm(x, (D,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], cfg.ln_eps))
        qkv = h @ sd[p + "attn.c_attn.weight"] + sd[p + "attn.c_attn.bias"]
        q, k, v = qkv.split(D, dim=2)
        q = q.view(B, S, H, dh).transpose(1, 2)
        k = rb(k.view(B, S, H, dh).transpose(1, 2))              # bf16 mode: the cache holds bf16 keys / values
        v = rb(v.view(B, S, H, dh).transpose(1, 2))
        if past is not None:
            k = torch.cat((past[i][0], k), dim=-2)                                    # transformers_gpt2.py:325-328
            v = torch.cat((past[i][1], v), dim=-2)
        new_kv.append((k, v))
        Tk = k.shape[-2]
        w = torch.matmul(q, k.transpose(-1, -2)) / (float(dh) ** 0.5)
        causal = torch.tril(torch.ones(Tk, Tk, dtype=torch.bool))[Tk - S:Tk, :Tk]
        w = torch.where(causal, w, torch.full([], fmin, dtype=w.dtype))
        if add_mask is not None:
            w = w + add_mask
        w = torch.softmax(w, dim=-1)
        a = rb(torch.matmul(w, v).transpose(1, 2).reshape(B, S, D))
        a = a @ sd[p + "attn.c_proj.weight"] + sd[p + "attn.c_proj.bias"]
        x = x + a
        h = rb(F.layer_norm(x, (D,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], cfg.ln_eps))
        h = rb(gelu_new(h @ sd[p + "mlp.c_fc.weight"] + sd[p + "mlp.c_fc.bias"]))
        h = h @ sd[p + "mlp.c_proj.weight"] + sd[p + "mlp.c_proj.bias"]
        x = x + h
    x = F.layer_norm(x, (D,), sd["gpt.ln_f.weight"], sd["gpt.ln_f.bias"], cfg.ln_eps)
    return x, new_kv


def lm_head(sd, cfg: GPTConfig, hidden: torch.Tensor) -> torch.Tensor:
    """lm_head = Sequential(final_norm, mel_head)  (model_v2.py:54)."""
    h = rb(F.layer_norm(hidden, (cfg.model_dim,), sd["final_norm.weight"], sd["final_norm.bias"], cfg.ln_eps))
    return F.linear(h, sd["mel_head.weight"], sd["mel_head.bias"])


# ----------------------------------------------------------------------------
# input assembly
# ----------------------------------------------------------------------------
def conds_latent_campplus(sd, style: torch.Tensor, emo_vec: torch.Tensor) -> torch.Tensor:
    """v2.5 conditioning: spk_emb_proj(style)+emo_vec, then 2 zero tokens (model_v2.py:754-755,768)."""
    spk = F.linear(style, sd["spk_emb_proj.weight"], sd["spk_emb_proj.bias"])
    if spk.ndim != 3:
        spk = spk.unsqueeze(0)
    return torch.cat((spk + emo_vec.unsqueeze(1), torch.zeros(spk.size(0), 2, spk.size(2))), 1)


def prepare_gpt_inputs(sd, cfg: GPTConfig, conds: torch.Tensor, text_inputs: torch.Tensor,
                       langs: Optional[torch.Tensor] = None):
    """model_v2.py:648-714.  conds (1|B, n_cond, D); text_inputs (B, L) int; langs (B,) or None.

    Returns fake_inputs (B,s+1) int64, embeds (B,s,D), attention_mask (B,s+1) int64.
    NOTE: the reference indexes `langs[i]`, so a batched call needs langs of
    shape (B,); a (1,)-shaped `langs` is broadcast here.
    """
    b, L = text_inputs.shape
    single = conds.shape[0] == 1
    n_cond = conds.shape[1]
    target_len = n_cond + L + 2
    embs, masks = [], []
    for i in range(b):
        row = text_inputs[i]
        valid = (row != cfg.stop_text_token) & (row != cfg.start_text_token)
        t = row[valid].long()
        t = F.pad(t, (1, 0), value=cfg.start_text_token)
        t = F.pad(t, (0, 1), value=cfg.stop_text_token)
        e = sd["text_embedding.weight"][t] + sd["text_pos_embedding.emb.weight"][: t.numel()]
        if langs is not None:
            li = langs[i] if langs.numel() > 1 else langs.reshape(-1)[0]
            e = e + sd["lang_embedding.weight"][li.long()]
        parts = [conds[0] if single else conds[i], e]
        m = torch.ones(target_len + 1, dtype=torch.long)
        padding = L + 2 - t.numel()
        if padding > 0:
            parts.insert(0, torch.zeros(padding, cfg.model_dim))
            m[:padding] = 0
        embs.append(torch.cat(parts))
        masks.append(m)
    embeds = torch.stack(embs)
    mask = torch.stack(masks)
    fake = torch.ones(b, target_len + 1, dtype=torch.long)
    fake[:, -1] = cfg.start_mel_token
    return fake, embeds, mask


# ----------------------------------------------------------------------------
# logits processors (order: generation_utils.py:900-901 then :1035-1044)
# ----------------------------------------------------------------------------
def repetition_penalty_(scores: torch.Tensor, input_ids: torch.Tensor, penalty: float) -> torch.Tensor:
    s = torch.gather(scores, 1, input_ids)
    s = torch.where(s < 0, s * penalty, s / penalty)
    return scores.scatter(1, input_ids, s)


def top_k_(scores: torch.Tensor, top_k: int, min_keep: int) -> torch.Tensor:
    k = min(max(top_k, min_keep), scores.size(-1))
    kth = torch.topk(scores, k)[0][..., -1, None]
    return scores.masked_fill(scores < kth, -float("inf"))


def top_p_(scores: torch.Tensor, top_p: float, min_keep: int) -> torch.Tensor:
    sorted_logits, sorted_idx = torch.sort(scores, descending=False)
    cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
    remove = cum <= (1 - top_p)
    remove[..., -min_keep:] = 0
    remove = remove.scatter(1, sorted_idx, remove)
    return scores.masked_fill(remove, -float("inf"))


def typical_(scores: torch.Tensor, mass: float, min_keep: int) -> torch.Tensor:
    """indextts/utils/typical_sampling.py:4-30."""
    normalized = torch.log_softmax(scores, dim=-1)
    p = torch.exp(normalized)
    ent = -(normalized * p).nansum(-1, keepdim=True)
    shifted = torch.abs((-normalized) - ent)
    sorted_scores, sorted_idx = torch.sort(shifted, descending=False)
    sorted_logits = scores.gather(-1, sorted_idx)
    cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
    last_ind = (cum < mass).sum(dim=1)
    last_ind[last_ind < 0] = 0
    remove = sorted_scores > sorted_scores.gather(1, last_ind.view(-1, 1))
    if min_keep > 1:
        remove[..., :min_keep] = 0
    remove = remove.scatter(1, sorted_idx, remove)
    return scores.masked_fill(remove, -float("inf"))


@dataclass
class GenParams:
    do_sample: bool = False
    num_beams: int = 1
    top_p: float = 0.8
    top_k: int = 30
    temperature: float = 0.8
    repetition_penalty: float = 10.0
    length_penalty: float = 0.0
    typical_sampling: bool = False
    typical_mass: float = 0.9
    max_generate_length: int = 100


def process_scores(scores: torch.Tensor, input_ids: torch.Tensor, gp: GenParams) -> torch.Tensor:
    min_keep = 2 if gp.num_beams > 1 else 1
    if gp.repetition_penalty is not None and gp.repetition_penalty != 1.0:
        scores = repetition_penalty_(scores, input_ids, gp.repetition_penalty)
    if gp.typical_sampling:                                     # user processor list (model_v2.py:793-799)
        scores = typical_(scores, gp.typical_mass, min_keep)
    if gp.do_sample:
        if gp.temperature is not None and gp.temperature != 1.0:
            scores = scores / gp.temperature
        if gp.top_k is not None and gp.top_k != 0:
            scores = top_k_(scores, gp.top_k, min_keep)
        if gp.top_p is not None and gp.top_p < 1.0:
            scores = top_p_(scores, gp.top_p, min_keep)
    return scores


def inverse_cdf_pick(probs: torch.Tensor, u: float) -> int:
    """Smallest index i with cumsum(probs)[i] > u * sum(probs); double accumulation, vocabulary order."""
    c = probs.double().cumsum(0)
    tgt = u * c[-1].item()
    idx = int(torch.searchsorted(c, torch.tensor(tgt, dtype=torch.float64), right=True).item())
    nz = torch.nonzero(probs > 0).flatten()
    idx = min(idx, int(nz[-1].item()))
    return idx


# ----------------------------------------------------------------------------
# inference forward with the reference's position rule
# ----------------------------------------------------------------------------
class InferenceModel:
    """GPT2InferenceModel (model_v2.py:46-198) with kv_cache=True/False semantics."""

    def __init__(self, sd, cfg: GPTConfig, kv_cache: bool = True):
        self.sd, self.cfg, self.kv_cache = sd, cfg, kv_cache
        self.cached_mel_emb: Optional[torch.Tensor] = None

    def store_mel_emb(self, e: torch.Tensor):
        self.cached_mel_emb = e

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, past: Optional[KV]):
        sd, cfg = self.sd, self.cfg
        mel_len = self.cached_mel_emb.shape[1]
        if not self.kv_cache:
            past = None
        if past is not None:
            input_ids = input_ids[:, -1:]                         # model_v2.py:96-97
        if input_ids.shape[1] != 1:
            ids = input_ids[:, mel_len:]
            e = sd["mel_embedding.weight"][ids] + sd["mel_pos_embedding.emb.weight"][: ids.shape[1]]
            pre = self.cached_mel_emb
            if pre.shape[0] != e.shape[0]:
                pre = pre.repeat_interleave(e.shape[0] // pre.shape[0], 0)       # model_v2.py:150-155
            emb = torch.cat([pre, e], dim=1)
        else:
            pos = attention_mask.shape[1] - mel_len                               # model_v2.py:158-161 (k+1 quirk)
            emb = sd["mel_embedding.weight"][input_ids] + sd["mel_pos_embedding.emb.weight"][pos]
        hidden, kv = gpt2_stack(sd, cfg, emb, attention_mask, past)
        return lm_head(sd, cfg, hidden), kv


# ----------------------------------------------------------------------------
# generate: greedy / sample (generation_utils.py:3123-3297)
# ----------------------------------------------------------------------------
def generate_sample(model: InferenceModel, inputs: torch.Tensor, attention_mask: torch.Tensor, gp: GenParams,
                    uniforms: Optional[torch.Tensor] = None, trace: Optional[dict] = None) -> torch.Tensor:
    cfg = model.cfg
    pad = eos = cfg.stop_mel_token
    input_ids = inputs.clone()
    B, cur_len = input_ids.shape
    max_length = cur_len + gp.max_generate_length                 # model_v2.py:792,800
    unfinished = torch.ones(B, dtype=torch.long)
    past = None
    step = 0
    while cur_len < max_length:
        logits, past = model.forward(input_ids, attention_mask, past)
        attention_mask = torch.cat([attention_mask, attention_mask.new_ones(B, 1)], dim=-1)   # :766-772
        next_logits = logits[:, -1, :].float().clone()
        if trace is not None:
            trace.setdefault("logits", []).append(next_logits.clone())
        scores = process_scores(next_logits, input_ids, gp)
        if trace is not None:
            trace.setdefault("scores", []).append(scores.clone())     # what the selection actually ranks (after the processors)
        if gp.do_sample:
            probs = torch.softmax(scores, dim=-1)
            nxt = torch.tensor([inverse_cdf_pick(probs[b], float(uniforms[step, b])) for b in range(B)])
        else:
            nxt = torch.argmax(scores, dim=-1)
        nxt = nxt * unfinished + pad * (1 - unfinished)           # :3256
        input_ids = torch.cat([input_ids, nxt[:, None]], dim=-1)
        cur_len += 1
        step += 1
        done = (input_ids[:, -1] == eos) | (cur_len >= max_length)
        unfinished = unfinished & (~done).long()
        if unfinished.max() == 0:
            break
    return input_ids


# ----------------------------------------------------------------------------
# beam search / beam sample (generation_utils.py:3325-3609 + transformers_beam_search.py)
# ----------------------------------------------------------------------------
class BeamHyps:
    def __init__(self, num_beams: int, length_penalty: float):
        self.num_beams, self.length_penalty = num_beams, length_penalty
        self.beams: List[Tuple[float, torch.Tensor]] = []
        self.worst_score = 1e9

    def add(self, hyp: torch.Tensor, sum_logprobs: float, generated_len: int):
        score = sum_logprobs / (generated_len ** self.length_penalty)
        if len(self.beams) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self.beams) > self.num_beams:
                order = sorted([(s, i) for i, (s, _) in enumerate(self.beams)])
                del self.beams[order[0][1]]
                self.worst_score = order[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, cur_len: int, prompt_len: int) -> bool:
        if len(self.beams) < self.num_beams:
            return False
        highest = best_sum_logprobs / (cur_len - prompt_len) ** self.length_penalty   # early_stopping=False
        return self.worst_score >= highest


class BeamScorer:
    def __init__(self, batch_size: int, num_beams: int, length_penalty: float):
        self.nb = num_beams
        self.hyps = [BeamHyps(num_beams, length_penalty) for _ in range(batch_size)]
        self.done = [False] * batch_size

    @property
    def is_done(self) -> bool:
        return all(self.done)

    def process(self, input_ids, next_scores, next_tokens, next_indices, pad, eos, prompt_len):
        cur_len = input_ids.shape[-1] + 1
        bs, nb = len(self.hyps), self.nb
        nbs = torch.zeros(bs, nb, dtype=next_scores.dtype)
        nbt = torch.zeros(bs, nb, dtype=next_tokens.dtype)
        nbi = torch.zeros(bs, nb, dtype=next_indices.dtype)
        for b in range(bs):
            if self.done[b]:
                nbs[b, :] = 0
                nbt[b, :] = pad
                nbi[b, :] = 0
                continue
            beam_idx = 0
            for rank, (tok, sc, ix) in enumerate(zip(next_tokens[b], next_scores[b], next_indices[b])):
                bb = b * nb + int(ix)
                if int(tok) == eos:
                    if rank >= nb:
                        continue
                    self.hyps[b].add(input_ids[bb].clone(), float(sc), cur_len - prompt_len)
                else:
                    nbs[b, beam_idx], nbt[b, beam_idx], nbi[b, beam_idx] = sc, tok, bb
                    beam_idx += 1
                if beam_idx == nb:
                    break
            assert beam_idx == nb, "not enough non-eos candidates"
            self.done[b] = self.done[b] or self.hyps[b].is_done(float(next_scores[b].max()), cur_len, prompt_len)
        return nbs.view(-1), nbt.view(-1), nbi.view(-1)

    def finalize(self, input_ids, final_scores, pad, eos, max_length, prompt_len):
        bs, nb = len(self.hyps), self.nb
        for b, hy in enumerate(self.hyps):
            if self.done[b]:
                continue
            for j in range(nb):
                bb = b * nb + j
                hy.add(input_ids[bb], float(final_scores[bb]), input_ids.shape[-1] - prompt_len)
        best = []
        for b in range(bs):
            srt = sorted(self.hyps[b].beams, key=lambda x: x[0])
            best.append(srt.pop()[1])
        lens = torch.tensor([len(h) for h in best])
        sent_max = min(int(lens.max()) + 1, max_length)
        out = torch.full((bs, sent_max), pad, dtype=torch.long) if int(lens.min()) != int(lens.max()) \
            else torch.zeros(bs, sent_max, dtype=torch.long)
        for i, h in enumerate(best):
            out[i, : lens[i]] = h
            if lens[i] < sent_max:
                out[i, lens[i]] = eos
        return out


def multinomial_wo_replacement(probs: torch.Tensor, n: int, us) -> List[int]:
    """Sequential draw-and-remove with one uniform per draw (distribution of torch.multinomial(replacement=False))."""
    p = probs.double().clone()
    out = []
    for j in range(n):
        i = inverse_cdf_pick(p, float(us[j]))
        out.append(i)
        p[i] = 0.0
    return out


def generate_beam(model: InferenceModel, inputs: torch.Tensor, attention_mask: torch.Tensor, gp: GenParams,
                  uniforms: Optional[torch.Tensor] = None) -> torch.Tensor:
    cfg = model.cfg
    pad = eos = cfg.stop_mel_token
    nb = gp.num_beams
    bs = inputs.shape[0]
    input_ids = inputs.repeat_interleave(nb, dim=0)               # _expand_inputs_for_generation :699-731
    attention_mask = attention_mask.repeat_interleave(nb, dim=0)
    prompt_len = cur_len = input_ids.shape[1]
    max_length = cur_len + gp.max_generate_length
    scorer = BeamScorer(bs, nb, gp.length_penalty)
    beam_scores = torch.zeros(bs, nb)
    beam_scores[:, 1:] = -1e9                                     # :3408-3410
    beam_scores = beam_scores.view(-1)
    past = None
    step = 0
    V = cfg.number_mel_codes
    n_keep = 2 * nb
    while True:
        logits, past = model.forward(input_ids, attention_mask, past)
        attention_mask = torch.cat([attention_mask, attention_mask.new_ones(bs * nb, 1)], dim=-1)
        nl = logits[:, -1, :].float().clone()
        ns = F.log_softmax(nl, dim=-1)
        ns = process_scores(ns, input_ids, gp)                    # processors act on LOG-PROBS here (:3475-3479)
        ns = ns + beam_scores[:, None]
        ns = ns.view(bs, nb * V)
        if gp.do_sample:
            probs = torch.softmax(ns, dim=-1)
            toks = torch.tensor([multinomial_wo_replacement(probs[b], n_keep, uniforms[step, b]) for b in range(bs)])
            sc = torch.gather(ns, -1, toks)
            sc, order = torch.sort(sc, descending=True, dim=1, stable=True)
            toks = torch.gather(toks, -1, order)
        else:
            sc, toks = torch.topk(ns, n_keep, dim=1, largest=True, sorted=True)
        idx = torch.div(toks, V, rounding_mode="floor")
        toks = toks % V
        beam_scores, beam_next, beam_idx = scorer.process(input_ids, sc, toks, idx, pad, eos, prompt_len)
        input_ids = torch.cat([input_ids[beam_idx, :], beam_next.unsqueeze(-1)], dim=-1)
        past = [(k.index_select(0, beam_idx), v.index_select(0, beam_idx)) for k, v in past]   # model_v2.py:200-213
        cur_len += 1
        step += 1
        if scorer.is_done or cur_len >= max_length:
            break
    return scorer.finalize(input_ids, beam_scores, pad, eos, max_length, prompt_len)


# ----------------------------------------------------------------------------
# driver: UnifiedVoice.inference_speech (model_v2.py:716-825), v2.5 / campplus conditioning
# ----------------------------------------------------------------------------
def inference_speech(sd, cfg: GPTConfig, conds_latent: torch.Tensor, text_inputs: torch.Tensor,
                     langs: Optional[torch.Tensor], gp: GenParams, uniforms: Optional[torch.Tensor] = None,
                     kv_cache: bool = True, trace: Optional[dict] = None) -> torch.Tensor:
    """Returns generated codes (B, n) = output[:, trunc_index:]."""
    fake, embeds, mask = prepare_gpt_inputs(sd, cfg, conds_latent, text_inputs, langs)
    model = InferenceModel(sd, cfg, kv_cache=kv_cache)
    model.store_mel_emb(embeds)
    trunc = fake.shape[1]
    if gp.num_beams > 1:
        out = generate_beam(model, fake, mask, gp, uniforms)
    else:
        out = generate_sample(model, fake, mask, gp, uniforms, trace)
    return out[:, trunc:]


# ----------------------------------------------------------------------------
# teacher-forced latent pass: UnifiedVoice.forward (model_v2.py:596-646)
# ----------------------------------------------------------------------------
def forward_latent(sd, cfg: GPTConfig, conds: torch.Tensor, text_inputs: torch.Tensor, text_lengths: torch.Tensor,
                   mel_codes: torch.Tensor, mel_lengths: torch.Tensor) -> torch.Tensor:
    """conds (B, n_cond, D) already assembled; returns final_norm hidden states of the mel positions (B, m, D).

    Rows are RIGHT-padded with stop tokens (set_text_padding / set_mel_padding, :500-526) and no attention mask is
    used (get_logits :534), exactly as the reference does.
    """
    text = text_inputs.clone().long()
    mel = mel_codes.clone().long()
    for b in range(text.shape[0]):
        if text_lengths[b] < text.shape[-1]:
            text[b, text_lengths[b]:] = cfg.stop_text_token
        if mel_lengths[b] < mel.shape[-1]:
            mel[b, mel_lengths[b]:] = cfg.stop_mel_token
    text = F.pad(text, (0, 1), value=cfg.stop_text_token)
    mel = F.pad(mel, (0, 1), value=cfg.stop_mel_token)
    text_in = F.pad(text, (1, 0), value=cfg.start_text_token)
    mel_in = F.pad(mel, (1, 0), value=cfg.start_mel_token)
    te = sd["text_embedding.weight"][text_in] + sd["text_pos_embedding.emb.weight"][: text_in.shape[1]]
    me = sd["mel_embedding.weight"][mel_in] + sd["mel_pos_embedding.emb.weight"][: mel_in.shape[1]]
    emb = torch.cat([conds, te, me], dim=1)
    hidden, _ = gpt2_stack(sd, cfg, emb, None, None)
    enc = hidden[:, conds.shape[1]:]
    enc = F.layer_norm(enc, (cfg.model_dim,), sd["final_norm.weight"], sd["final_norm.bias"], cfg.ln_eps)
    mel_lat = enc[:, -mel_in.shape[1]:]
    return mel_lat[:, :-2]


def forward_latent_v1(sd, cfg: GPTConfig, conds: torch.Tensor, text_inputs: torch.Tensor, text_lengths: torch.Tensor,
                      mel_codes: torch.Tensor, wav_lengths: torch.Tensor, mel_length_compression: int = 1024) -> torch.Tensor:
    """IndexTTS-1/1.5 `UnifiedVoice.forward(..., return_latent=True)` (indextts/gpt/model.py:526-590): the same pass as
    `forward_latent`, with the mel lengths derived from the waveform lengths, `ceil(wav/compression) + 1` (:557), and
    `conds` = the (B, 32, D) output of get_conditioning."""
    mel_lengths = torch.ceil(wav_lengths / mel_length_compression).long() + 1
    return forward_latent(sd, cfg, conds, text_inputs, text_lengths, mel_codes, mel_lengths)
