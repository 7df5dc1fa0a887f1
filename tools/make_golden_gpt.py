#!/usr/bin/env python3
"""Mint GPT golden vectors by running the REFERENCE's own code (build container only).

What runs here is reference source, read from /root/reference at run time (never copied into this repo):
  * `GPT2InferenceModel`, `LearnedPositionEmbeddings`, `null_position_embeddings` and the `UnifiedVoice`
    methods `prepare_gpt_inputs`, `inference_speech`, `forward`, `get_logits`, `set_text_padding`,
    `set_mel_padding`, `build_aligned_inputs_and_targets` are extracted from `indextts/gpt/model_v2.py` by
    AST and exec'd (the module itself is not importable under transformers 5.15 -- tools/ref_shim.py);
  * `generate()` / `_sample` / `_beam_search` / `_get_logits_processor` are the reference's vendored
    `indextts/gpt/transformers_generation_utils.py` + `transformers_beam_search.py` (tools/ref_shim.py);
  * the transformer blocks are the installed HF `GPT2Model(attn_implementation="eager")` with `wpe` nulled
    exactly as `build_hf_gpt_transformer` does (model_v2.py:259-279) behind a thin adapter that converts the
    legacy KV tuples the vendored mixin carries to/from the 5.15 cache object;
  * logits processors are the installed `transformers.generation.logits_process` classes.
`torch.multinomial` is replaced (only inside sampled runs) by inverse-CDF draws from a stored uniform stream,
since the RNG stream itself cannot be reproduced on a device; that stream is part of the fixture.

Outputs tests/golden/gpt_*.npz (inputs, uniforms, reference token ids / latents; weights are regenerated
from the seed by oracle.gpt_oracle.synth_weights) and prints oracle-vs-reference agreement.
"""
import ast
import functools
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ref_shim  # noqa: E402

gu, _log = ref_shim.load_ref_generation_utils()

from transformers import GPT2Config, GPT2Model, GenerationConfig, LogitsProcessorList  # noqa: E402
from transformers.cache_utils import DynamicCache  # noqa: E402
import importlib.util  # noqa: E402
_sp = importlib.util.spec_from_file_location("ref_typical_sampling", os.path.join(ref_shim.REF, "indextts/utils/typical_sampling.py"))
_ref_typical = importlib.util.module_from_spec(_sp)
_sp.loader.exec_module(_ref_typical)                       # the reference's own TypicalLogitsWarper (typical_sampling.py:4-30)
HFTypical = _ref_typical.TypicalLogitsWarper
from transformers.modeling_outputs import CausalLMOutputWithCrossAttentions  # noqa: E402

from oracle import gpt_oracle as G  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
MODEL_V2 = "/root/reference/indextts/gpt/model_v2.py"
MODEL_V1 = "/root/reference/indextts/gpt/model.py"

LEGACY_GENCFG = {"return_legacy_cache", "forced_decoder_ids"}     # attributes 4.52's GenerationConfig had
# transformers 4.52.1 GenerationConfig defaults (5.15 initialises every field to None)
DEFAULTS_4_52 = dict(
    max_length=20, min_length=0, early_stopping=False, do_sample=False, num_beams=1, num_beam_groups=1,
    use_cache=True, temperature=1.0, top_k=50, top_p=1.0, typical_p=1.0, epsilon_cutoff=0.0, eta_cutoff=0.0,
    diversity_penalty=0.0, repetition_penalty=1.0, encoder_repetition_penalty=1.0, length_penalty=1.0,
    no_repeat_ngram_size=0, encoder_no_repeat_ngram_size=0, renormalize_logits=False, remove_invalid_values=False,
    token_healing=False, num_return_sequences=1, output_attentions=False, output_hidden_states=False,
    output_scores=False, return_dict_in_generate=False)


class LegacyGenerationConfig(GenerationConfig):
    def __getattr__(self, k):
        if k in LEGACY_GENCFG:
            return None
        raise AttributeError(k)


class HarnessPreTrainedModel(nn.Module, gu.GenerationMixin):
    """Stands in for the vendored GPT2PreTrainedModel base: just enough for the vendored generate()."""
    main_input_name = "input_ids"
    _supports_cache_class = False
    _is_stateful = False

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.generation_config = LegacyGenerationConfig(**DEFAULTS_4_52)

    @property
    def device(self):
        return torch.device("cpu")

    def can_generate(self):
        return True


class TransformerAdapter(nn.Module):
    """Installed HF GPT2Model behind the 4.52-era call signature (legacy KV tuples in/out)."""

    def __init__(self, gpt):
        super().__init__()
        self.gpt = gpt

    def forward(self, inputs_embeds=None, past_key_values=None, attention_mask=None, use_cache=None,
                return_dict=None, **unused):
        cache = None
        if past_key_values is not None:
            cache = DynamicCache()
            for li, (k, v) in enumerate(past_key_values):
                cache.update(k, v, li)
        elif use_cache:
            cache = DynamicCache()
        out = self.gpt(inputs_embeds=inputs_embeds, past_key_values=cache, attention_mask=attention_mask,
                       use_cache=bool(use_cache), return_dict=True)
        pkv = None
        if out.past_key_values is not None:
            c = out.past_key_values
            pkv = tuple((c.layers[i].keys, c.layers[i].values) for i in range(len(c.layers)))
        out.past_key_values = pkv
        return out


def extract(names_top, methods_of=None, method_names=(), path=None):
    """Pull class/function definitions (and selected methods of a class) out of model_v2.py / model.py as source."""
    src = open(path or MODEL_V2).read()
    tree = ast.parse(src)
    out = {}
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in names_top:
            out[node.name] = ast.get_source_segment(src, node)
        if isinstance(node, ast.ClassDef) and node.name == methods_of:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name in method_names:
                    import textwrap
                    out[sub.name] = textwrap.dedent(ast.get_source_segment(src, sub, padded=True))
    return out


def build_reference(sd, cfg: G.GPTConfig, kv_cache=True, v1=False):
    """v1=True: the IndexTTS-1/1.5 classes of indextts/gpt/model.py (no language embedding, 32-token conditioning
    latent from get_conditioning -- stubbed to return the latent it is handed, the encoder is outside the hot path)."""
    MODEL = MODEL_V1 if v1 else MODEL_V2
    ns = dict(torch=torch, nn=nn, F=F, functools=functools, GPT2PreTrainedModel=HarnessPreTrainedModel,
              CausalLMOutputWithCrossAttentions=CausalLMOutputWithCrossAttentions,
              LogitsProcessorList=LogitsProcessorList, TypicalLogitsWarper=HFTypical,
              get_device_map=None, assert_device_map=None)
    pieces = extract({"GPT2InferenceModel", "LearnedPositionEmbeddings", "null_position_embeddings"},
                     "UnifiedVoice", ("prepare_gpt_inputs", "inference_speech", "forward", "get_logits",
                                      "set_text_padding", "set_mel_padding", "build_aligned_inputs_and_targets"),
                     path=MODEL)
    for name in ("null_position_embeddings", "LearnedPositionEmbeddings", "GPT2InferenceModel"):
        exec(compile(pieces[name], MODEL + ":" + name, "exec"), ns)

    class RefUnifiedVoice(nn.Module):
        pass

    for name in ("prepare_gpt_inputs", "inference_speech", "forward", "get_logits", "set_text_padding",
                 "set_mel_padding", "build_aligned_inputs_and_targets"):
        exec(compile(pieces[name], MODEL + ":" + name, "exec"), ns)
        setattr(RefUnifiedVoice, name, ns[name])

    D = cfg.model_dim
    uv = RefUnifiedVoice()
    # attributes UnifiedVoice.__init__ sets (model_v2.py:334-410)
    uv.start_text_token, uv.stop_text_token = cfg.start_text_token, cfg.stop_text_token
    uv.start_mel_token, uv.stop_mel_token = cfg.start_mel_token, cfg.stop_mel_token
    uv.max_mel_tokens, uv.max_text_tokens = cfg.max_mel_tokens, cfg.max_text_tokens
    uv.spk_cond_mode = "campplus"
    uv.accel_engine = None
    uv.mel_length_compression = 1024
    if v1:
        uv.get_conditioning = lambda latent, lengths=None: latent        # (b, 32, D) handed in directly
    uv.spk_emb_proj = nn.Linear(192, D)
    uv.text_embedding = nn.Embedding(cfg.number_text_tokens * cfg.types + 1, D)
    uv.lang_embedding = nn.Embedding(cfg.n_langs, D)
    uv.mel_embedding = nn.Embedding(cfg.number_mel_codes, D)
    uv.mel_pos_embedding = ns["LearnedPositionEmbeddings"](cfg.n_mel_pos, D)
    uv.text_pos_embedding = ns["LearnedPositionEmbeddings"](cfg.n_text_pos, D)
    uv.final_norm = nn.LayerNorm(D)
    uv.mel_head = nn.Linear(D, cfg.number_mel_codes)
    uv.text_head = nn.Linear(D, cfg.number_text_tokens * cfg.types + 1)
    hf_cfg = GPT2Config(vocab_size=256, n_positions=cfg.n_mel_pos + cfg.n_text_pos, n_embd=D, n_layer=cfg.layers,
                        n_head=cfg.heads, attn_implementation="eager")
    gpt = GPT2Model(hf_cfg)
    del gpt.wpe
    gpt.wpe = functools.partial(ns["null_position_embeddings"], dim=D)        # model_v2.py:273-275
    del gpt.wte
    uv.gpt = TransformerAdapter(gpt)
    missing, unexpected = uv.load_state_dict({("gpt.gpt." + k[4:] if k.startswith("gpt.") else k): v
                                              for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all(m.startswith("text_head") for m in missing), missing
    gen_cfg = GPT2Config(vocab_size=cfg.number_mel_codes, n_positions=cfg.max_mel_tokens + cfg.max_text_tokens + 2,
                         n_embd=D, n_layer=cfg.layers, n_head=cfg.heads, use_cache=True)
    # post_init_gpt2_config (model_v2.py:466-474)
    uv.inference_model = ns["GPT2InferenceModel"](gen_cfg, uv.gpt, uv.mel_pos_embedding, uv.mel_embedding,
                                                   uv.final_norm, uv.mel_head, kv_cache=kv_cache).eval()
    return uv.eval()


class UniformMultinomial:
    """Context manager: torch.multinomial -> inverse-CDF draws from a stored uniform stream."""

    def __init__(self, uniforms: torch.Tensor):
        self.u = uniforms
        self.step = 0

    def __call__(self, probs, num_samples=1, replacement=False, **kw):
        rows = []
        for b in range(probs.shape[0]):
            us = self.u[self.step, b]
            if num_samples == 1:
                rows.append([G.inverse_cdf_pick(probs[b], float(us.reshape(-1)[0]))])
            else:
                rows.append(G.multinomial_wo_replacement(probs[b], num_samples, us))
        self.step += 1
        return torch.tensor(rows, dtype=torch.long)

    def __enter__(self):
        self._orig = torch.multinomial
        torch.multinomial = self
        return self

    def __exit__(self, *a):
        torch.multinomial = self._orig


def ragged_text(g, B, L, n_text, lens):
    t = torch.randint(2, n_text, (B, L), generator=g)
    for b, n in enumerate(lens):
        t[b, n:] = 1                      # stop_text_token padding, stripped + left-padded by prepare_gpt_inputs
    return t


def main():
    cases = {
        # tag: (cfg kwargs, seed, B, L, lens, gen kwargs, kv_cache, eos_bias)
        "greedy": (dict(layers=3, model_dim=128, heads=2), 21, 3, 10, [10, 7, 4],
                   dict(do_sample=False, num_beams=1, repetition_penalty=10.0), True, 2.2),
        "greedy_nokv": (dict(layers=2, model_dim=128, heads=2), 22, 2, 8, [8, 5],
                        dict(do_sample=False, num_beams=1, repetition_penalty=10.0), False, 2.2),
        "sample": (dict(layers=3, model_dim=128, heads=2), 23, 3, 10, [10, 6, 9],
                   dict(do_sample=True, num_beams=1, top_p=0.8, top_k=30, temperature=0.8, repetition_penalty=10.0),
                   True, 1.9),
        "beam": (dict(layers=2, model_dim=128, heads=2), 24, 2, 9, [9, 6],
                 dict(do_sample=False, num_beams=3, repetition_penalty=10.0, length_penalty=0.0), True, 1.2),
        "beam_sample": (dict(layers=2, model_dim=128, heads=2), 25, 3, 9, [9, 5, 7],
                        dict(do_sample=True, num_beams=3, top_p=0.8, top_k=30, temperature=0.8,
                             repetition_penalty=10.0, length_penalty=0.0), True, 1.5),
        "greedy_mid": (dict(layers=4, model_dim=256, heads=4), 26, 4, 16, [16, 12, 9, 16],
                       dict(do_sample=False, num_beams=1, repetition_penalty=10.0), True, 2.5),
        # typical sampling (model_v2.py:794-799 -> indextts/utils/typical_sampling.py) in the three loop modes
        "typical_sample": (dict(layers=2, model_dim=128, heads=2), 27, 3, 9, [9, 6, 8],
                           dict(do_sample=True, num_beams=1, top_p=0.8, top_k=30, temperature=0.8, repetition_penalty=10.0,
                                typical_sampling=True, typical_mass=0.9), True, 1.9),
        "typical_greedy": (dict(layers=2, model_dim=128, heads=2), 28, 3, 9, [9, 4, 7],
                           dict(do_sample=False, num_beams=1, repetition_penalty=10.0, typical_sampling=True,
                                typical_mass=0.2), True, 2.2),
        "typical_beam_sample": (dict(layers=2, model_dim=128, heads=2), 29, 2, 9, [9, 6],
                                dict(do_sample=True, num_beams=3, top_p=0.8, top_k=30, temperature=0.8,
                                     repetition_penalty=10.0, length_penalty=0.0, typical_sampling=True,
                                     typical_mass=0.5), True, 1.5),
    }
    only = sys.argv[1] if len(sys.argv) > 1 else None          # substring filter: regenerate a subset of the cases
    max_gen = 28
    for tag, (ck, seed, B, L, lens, gk, kv, eos_bias) in cases.items():
        if only and only not in tag:
            continue
        cfg = G.GPTConfig(max_text_tokens=40, max_mel_tokens=60, number_text_tokens=200, **ck)
        sd = G.synth_weights(cfg, seed=seed)
        sd["mel_head.bias"][cfg.stop_mel_token] += eos_bias          # make EOS reachable at ragged steps
        g = torch.Generator().manual_seed(seed + 100)
        text = ragged_text(g, B, L, cfg.number_text_tokens, lens)
        style = torch.randn(1, 192, generator=g)
        emo_vec = torch.randn(1, cfg.model_dim, generator=g) * 0.1
        langs = torch.randint(0, cfg.n_langs, (B,), generator=g)
        nb = gk.get("num_beams", 1)
        uniforms = torch.rand(max_gen + 2, B, 2 * nb if nb > 1 else 1, generator=g, dtype=torch.float64)
        uv = build_reference(sd, cfg, kv_cache=kv)
        with torch.no_grad(), UniformMultinomial(uniforms):
            codes, spk_lat = uv.inference_speech(torch.zeros(1, 4, 2), text, langs=langs, emo_vec=emo_vec,
                                                 campplus_embedding=style, max_generate_length=max_gen, **gk)
        # oracle
        gp = G.GenParams(max_generate_length=max_gen, **gk)
        conds = G.conds_latent_campplus(sd, style, emo_vec)
        u = uniforms if nb > 1 else uniforms[..., 0]
        with torch.no_grad():
            oc = G.inference_speech(sd, cfg, conds, text, langs, gp, uniforms=u, kv_cache=kv)
        same = codes.shape == oc.shape and bool((codes == oc).all())
        eos_at = [(int((r == cfg.stop_mel_token).nonzero()[0]) if (r == cfg.stop_mel_token).any() else -1) for r in codes]
        print(f"{tag}: ref codes {tuple(codes.shape)} eos_at={eos_at} oracle==reference: {same}")
        if not same:
            print(codes, oc)
        np.savez_compressed(
            os.path.join(GOLD, f"gpt_{tag}.npz"), text=text.numpy(), style=style.numpy(), emo_vec=emo_vec.numpy(),
            langs=langs.numpy(), uniforms=uniforms.numpy(), codes=codes.numpy(), seed=np.int64(seed),
            eos_bias=np.float64(eos_bias), kv_cache=np.bool_(kv), max_gen=np.int64(max_gen),
            cfg=np.array([cfg.layers, cfg.model_dim, cfg.heads, cfg.max_text_tokens, cfg.max_mel_tokens,
                          cfg.number_text_tokens]),
            gen=np.array([int(gk.get("do_sample", False)), nb, gk.get("top_p", 1.0), gk.get("top_k", 0),
                          gk.get("temperature", 1.0), gk.get("repetition_penalty", 1.0),
                          gk.get("length_penalty", 1.0), int(gk.get("typical_sampling", False)),
                          gk.get("typical_mass", 0.9)], dtype=np.float64))

        if tag == "greedy":
            # teacher-forced latent pass (UnifiedVoice.forward, model_v2.py:596-646) on the generated codes
            tl = torch.tensor(lens)
            ml = torch.tensor([max(2, (eos_at[b] if eos_at[b] >= 0 else codes.shape[1]) - 0) for b in range(B)])
            mel_codes = codes[:, : int(ml.max())].clone()
            condsB = conds.repeat(B, 1, 1)
            with torch.no_grad():
                # do_spk_cond=False: speech_conditioning_latent is used as-is; emo_vec given
                spk = F.linear(style, sd["spk_emb_proj.weight"], sd["spk_emb_proj.bias"]).unsqueeze(0).repeat(B, 1, 1)
                lat_ref = uv.forward(spk, text.clone(), tl, mel_codes.clone(), ml, None, emo_vec=emo_vec.repeat(B, 1),
                                     do_spk_cond=False)
                lat_o = G.forward_latent(sd, cfg, condsB, text, tl, mel_codes, ml)
            d = (lat_ref - lat_o).abs().max().item()
            print(f"  latent pass: ref {tuple(lat_ref.shape)} oracle max|d| = {d:.3e}")
            np.savez_compressed(os.path.join(GOLD, "gpt_latent.npz"), text=text.numpy(), text_lens=tl.numpy(),
                                mel_codes=mel_codes.numpy(), mel_lens=ml.numpy(), style=style.numpy(),
                                emo_vec=emo_vec.numpy(), latent=lat_ref.numpy().astype(np.float32),
                                seed=np.int64(seed), eos_bias=np.float64(eos_bias),
                                cfg=np.array([cfg.layers, cfg.model_dim, cfg.heads, cfg.max_text_tokens,
                                              cfg.max_mel_tokens, cfg.number_text_tokens]))

    if only is not None and only == "fullsize":            # minutes of CPU: only on request (`make_golden_gpt.py fullsize`)
        make_fullsize()
    if only is not None and only == "fullsize_beam":       # `make_golden_gpt.py fullsize_beam`
        make_fullsize_beam()
    if only is None or "v1" in only:
        make_v1()
    if only is None or "bf16" in only:
        make_bf16()
    if only is None or "v2" in only:
        make_v2()


def make_fullsize():
    """The benchmarked GPT at the benchmarked context: the full-size stack (24 x 1280 x 20 heads) decoding
    560 greedy tokens from 128-token texts, i.e. context 134 + 560 = 694 -- BASELINE configs[2]'s shape, two rows (one ragged).
    The REFERENCE's own classes run it here on CPU (minutes); the fixture holds their ids plus the oracle's top-2 margins of the
    processed scores per step (the oracle's ids must equal the reference's), so the GPU test needs no CPU decode."""
    cfg = G.GPTConfig(max_text_tokens=140, max_mel_tokens=600)
    seed, B, L, n = 1234, 2, 128, 560
    sd = G.synth_weights(cfg, seed=seed)
    sd["mel_head.bias"][cfg.stop_mel_token] -= 1e4                # fixed-length decode (bench.py does the same)
    g = torch.Generator().manual_seed(692)          # text seed chosen for a comfortable minimum greedy margin (1.1e-3)
    lens = [128, 97]
    text = ragged_text(g, B, L, cfg.number_text_tokens, lens)
    style = torch.randn(1, 192, generator=g)
    emo_vec = torch.randn(1, cfg.model_dim, generator=g) * 0.1
    langs = torch.randint(0, cfg.n_langs, (B,), generator=g)
    gk = dict(do_sample=False, num_beams=1, repetition_penalty=10.0)
    import time
    t0 = time.time()
    uv = build_reference(sd, cfg, kv_cache=True)
    with torch.no_grad():
        codes, _ = uv.inference_speech(torch.zeros(1, 4, 2), text, langs=langs, emo_vec=emo_vec, campplus_embedding=style,
                                       max_generate_length=n, **gk)
    t1 = time.time()
    trace = {}
    with torch.no_grad():
        oc = G.inference_speech(sd, cfg, G.conds_latent_campplus(sd, style, emo_vec), text, langs,
                                G.GenParams(max_generate_length=n, **gk), kv_cache=True, trace=trace)
    t2 = time.time()
    same = codes.shape == oc.shape and bool((codes == oc).all())
    margins = np.stack([(lambda t: (t[:, 0] - t[:, 1]).numpy())(torch.topk(l, 2, dim=-1).values) for l in trace["scores"]], 1)
    print(f"fullsize: reference ids {tuple(codes.shape)} in {t1 - t0:.0f}s, oracle in {t2 - t1:.0f}s, oracle==reference: {same}; "
          f"min top-2 margin of the processed scores {margins.min():.3e} (row/step {np.unravel_index(margins.argmin(), margins.shape)})")
    assert same
    np.savez_compressed(os.path.join(GOLD, "gpt_fullsize_ctx694.npz"), text=text.numpy(), lens=np.array(lens), style=style.numpy(),
                        emo_vec=emo_vec.numpy(), langs=langs.numpy(), codes=codes.numpy(), margins=margins.astype(np.float32),
                        seed=np.int64(seed), n=np.int64(n),
                        cfg=np.array([cfg.layers, cfg.model_dim, cfg.heads, cfg.max_text_tokens, cfg.max_mel_tokens, cfg.number_text_tokens]))


def make_fullsize_beam():
    """The reference's DEFAULT generation mode (infer_v2_5.py:732-740: do_sample, top_p 0.8, top_k 30, temperature 0.8, num_beams 3,
    repetition_penalty 10, length_penalty 0) on the full-size stack (24 x 1280 x 20 heads): two utterances of 64 / 47 text tokens
    (BASELINE configs[1]'s text length), 3 beams each, 200 beam-sample steps, fixed-length decode.  The REFERENCE's own classes (vendored
    GenerationMixin._beam_search + BeamSearchScorer over HF GPT2Model) run it here on CPU with the explicit uniform stream; the oracle's ids
    must equal theirs; the fixture holds the ids and the uniforms (beam fixtures were small-model only)."""
    cfg = G.GPTConfig(max_text_tokens=80, max_mel_tokens=260)
    seed, B, L, n, nb = 4321, 2, 64, 200, 3
    sd = G.synth_weights(cfg, seed=seed)
    sd["mel_head.bias"][cfg.stop_mel_token] -= 1e4                # fixed-length decode (bench.py does the same)
    g = torch.Generator().manual_seed(seed + 100)
    lens = [64, 47]
    text = ragged_text(g, B, L, cfg.number_text_tokens, lens)
    style = torch.randn(1, 192, generator=g)
    emo_vec = torch.randn(1, cfg.model_dim, generator=g) * 0.1
    langs = torch.randint(0, cfg.n_langs, (B,), generator=g)
    uniforms = torch.rand(n + 2, B, 2 * nb, generator=g, dtype=torch.float64)
    gk = dict(do_sample=True, num_beams=nb, top_p=0.8, top_k=30, temperature=0.8, repetition_penalty=10.0, length_penalty=0.0)
    import time
    t0 = time.time()
    uv = build_reference(sd, cfg, kv_cache=True)
    with torch.no_grad(), UniformMultinomial(uniforms):
        codes, _ = uv.inference_speech(torch.zeros(1, 4, 2), text, langs=langs, emo_vec=emo_vec, campplus_embedding=style,
                                       max_generate_length=n, **gk)
    t1 = time.time()
    with torch.no_grad():
        oc = G.inference_speech(sd, cfg, G.conds_latent_campplus(sd, style, emo_vec), text, langs,
                                G.GenParams(max_generate_length=n, **gk), uniforms=uniforms, kv_cache=True)
    t2 = time.time()
    same = codes.shape == oc.shape and bool((codes == oc).all())
    print(f"fullsize_beam: reference ids {tuple(codes.shape)} in {t1 - t0:.0f}s, oracle in {t2 - t1:.0f}s, oracle==reference: {same}")
    assert same
    np.savez_compressed(os.path.join(GOLD, "gpt_fullsize_beam3.npz"), text=text.numpy(), lens=np.array(lens), style=style.numpy(),
                        emo_vec=emo_vec.numpy(), langs=langs.numpy(), uniforms=uniforms.numpy(), codes=codes.numpy(),
                        seed=np.int64(seed), n=np.int64(n), nb=np.int64(nb),
                        gen=np.array([1, nb, 0.8, 30, 0.8, 10.0, 0.0], dtype=np.float64),
                        cfg=np.array([cfg.layers, cfg.model_dim, cfg.heads, cfg.max_text_tokens, cfg.max_mel_tokens, cfg.number_text_tokens]))


def make_v1():
    """BASELINE configs[0] shape in miniature: IndexTTS-1/1.5 `UnifiedVoice` (indextts/gpt/model.py), greedy decode with
    kv_cache=False (the reference's CPU setting, infer.py:99-101), 32-token conditioning latent, then the teacher-forced
    latent pass `self.gpt(..., return_latent=True)` (infer.py:638-643) on the generated codes."""
    seed, B, L, lens, max_gen, eos_bias = 33, 2, 9, [9, 6], 24, 1.7
    cfg = G.GPTConfig(layers=2, model_dim=128, heads=2, max_text_tokens=40, max_mel_tokens=60, number_text_tokens=200)
    sd = G.synth_weights(cfg, seed=seed)
    sd["mel_head.bias"][cfg.stop_mel_token] += eos_bias
    g = torch.Generator().manual_seed(seed + 100)
    text = ragged_text(g, B, L, cfg.number_text_tokens, lens)
    conds = torch.randn(1, 32, cfg.model_dim, generator=g) * 0.3
    gk = dict(do_sample=False, num_beams=1, repetition_penalty=10.0)
    uv = build_reference(sd, cfg, kv_cache=False, v1=True)
    with torch.no_grad():
        codes = uv.inference_speech(conds, text, cond_mel_lengths=torch.tensor([7]), max_generate_length=max_gen, **gk)
        oc = G.inference_speech(sd, cfg, conds, text, None, G.GenParams(max_generate_length=max_gen, **gk), kv_cache=False)
    same = codes.shape == oc.shape and bool((codes == oc).all())
    eos_at = [(int((r == cfg.stop_mel_token).nonzero()[0]) if (r == cfg.stop_mel_token).any() else -1) for r in codes]
    print(f"v1_greedy_nokv: ref codes {tuple(codes.shape)} eos_at={eos_at} oracle==reference: {same}")
    code_lens = torch.tensor([(e if e >= 0 else codes.shape[1]) for e in eos_at])
    code_lens[1] = min(int(code_lens[1]), 12)              # ragged lengths for set_mel_padding (model.py:439-451)
    mel_codes = codes[:, : int(code_lens.max())].clone()
    tl = torch.tensor(lens)
    with torch.no_grad():
        lat_ref = uv.forward(conds.repeat(B, 1, 1), text.clone(), tl, mel_codes.clone(), code_lens * uv.mel_length_compression,
                             cond_mel_lengths=torch.tensor([7]), return_latent=True, clip_inputs=False)
        lat_o = G.forward_latent_v1(sd, cfg, conds.repeat(B, 1, 1), text, tl, mel_codes, code_lens * 1024)
    print(f"  v1 latent pass: ref {tuple(lat_ref.shape)} oracle max|d| = {(lat_ref - lat_o).abs().max().item():.3e}")
    np.savez_compressed(os.path.join(GOLD, "gpt_v1.npz"), text=text.numpy(), text_lens=tl.numpy(), conds=conds.numpy(),
                        codes=codes.numpy(), code_lens=code_lens.numpy(), mel_codes=mel_codes.numpy(),
                        latent=lat_ref.numpy().astype(np.float32), seed=np.int64(seed), eos_bias=np.float64(eos_bias),
                        max_gen=np.int64(max_gen),
                        cfg=np.array([cfg.layers, cfg.model_dim, cfg.heads, cfg.max_text_tokens, cfg.max_mel_tokens,
                                      cfg.number_text_tokens]))


def make_v2():
    """BASELINE configs[3] in miniature: IndexTTS-2 `UnifiedVoice` with the reference DEFAULT conditioning mode
    (spk_cond_mode="conformer", indextts/infer_v2.py:98): 34 conditioning tokens = 32 speaker latents (get_conditioning is
    stubbed to return the latent it is handed -- the Conformer + Perceiver encoder is outside the hot path) + emo_vec, then
    speed_emb(1), speed_emb(0) (model_v2.py:767-773); no language embedding although a third positional argument is passed
    (infer_v2.py:584 hands `emo_cond_emb` where `langs` sits; :680 ignores it outside campplus mode).  Greedy decode, then the
    teacher-forced latent pass `self.gpt(...)` of infer_v2.py:636-651 with use_speed = 0."""
    seed, B, L, lens, max_gen, eos_bias = 47, 3, 12, [12, 7, 10], 20, 1.6
    cfg = G.GPTConfig(layers=3, model_dim=128, heads=2, max_text_tokens=40, max_mel_tokens=60, number_text_tokens=200)
    sd = G.synth_weights(cfg, seed=seed)
    sd["mel_head.bias"][cfg.stop_mel_token] += eos_bias
    g = torch.Generator().manual_seed(seed + 100)
    text = ragged_text(g, B, L, cfg.number_text_tokens, lens)
    spk_lat = torch.randn(1, 32, cfg.model_dim, generator=g) * 0.3
    emo_vec = torch.randn(1, cfg.model_dim, generator=g) * 0.1
    speed = torch.randn(2, cfg.model_dim, generator=g) * 0.3
    gk = dict(do_sample=False, num_beams=1, repetition_penalty=10.0)
    uv = build_reference(sd, cfg, kv_cache=True)
    uv.spk_cond_mode = "conformer"
    uv.speed_emb = nn.Embedding(2, cfg.model_dim)
    uv.speed_emb.weight.data.copy_(speed)
    uv.get_conditioning = lambda x, lengths=None: spk_lat.repeat(B, 1, 1)       # (b, 32, D): the reference concatenates per-row speed embeddings
    with torch.no_grad():
        # third positional = `langs` (what infer_v2.py passes there is the emotion feature tensor; ignored in this mode)
        codes, lat_out = uv.inference_speech(torch.zeros(1, 4, 2), text, torch.zeros(1, 4, 2), emo_vec=emo_vec,
                                             cond_lengths=torch.tensor([2]), emo_cond_lengths=torch.tensor([2]),
                                             max_generate_length=max_gen, **gk)
        conds = torch.cat((spk_lat + emo_vec.unsqueeze(1), speed[1][None, None], speed[0][None, None]), 1)
        oc = G.inference_speech(sd, cfg, conds, text, None, G.GenParams(max_generate_length=max_gen, **gk))
    same = codes.shape == oc.shape and bool((codes == oc).all())
    eos_at = [(int((r == cfg.stop_mel_token).nonzero()[0]) if (r == cfg.stop_mel_token).any() else -1) for r in codes]
    print(f"v2_greedy (34 conditioning tokens): ref codes {tuple(codes.shape)} eos_at={eos_at} oracle==reference: {same}; "
          f"returned latent {tuple(lat_out.shape)}")
    tl = torch.tensor(lens)
    ml = torch.tensor([max(2, (e if e >= 0 else codes.shape[1])) for e in eos_at])
    mel_codes = codes[:, : int(ml.max())].clone()
    with torch.no_grad():
        lat_ref = uv.forward(spk_lat.repeat(B, 1, 1), text.clone(), tl, mel_codes.clone(), ml, None, emo_vec=emo_vec.repeat(B, 1),
                             use_speed=torch.zeros(B).long(), do_spk_cond=False)
        lat_o = G.forward_latent(sd, cfg, conds.repeat(B, 1, 1), text, tl, mel_codes, ml)
    print(f"  v2 latent pass: ref {tuple(lat_ref.shape)} oracle max|d| = {(lat_ref - lat_o).abs().max().item():.3e}")
    np.savez_compressed(os.path.join(GOLD, "gpt_v2.npz"), text=text.numpy(), text_lens=tl.numpy(), spk_latent=spk_lat.numpy(),
                        emo_vec=emo_vec.numpy(), speed_emb=speed.numpy(), codes=codes.numpy(), mel_codes=mel_codes.numpy(),
                        mel_lens=ml.numpy(), latent=lat_ref.numpy().astype(np.float32), seed=np.int64(seed),
                        eos_bias=np.float64(eos_bias), max_gen=np.int64(max_gen),
                        cfg=np.array([cfg.layers, cfg.model_dim, cfg.heads, cfg.max_text_tokens, cfg.max_mel_tokens,
                                      cfg.number_text_tokens]))


def make_bf16():
    """The reference's OWN bf16 mode on the CPU: `self.gpt.eval().bfloat16()` (indextts/infer_v2_5.py:143-146) under
    `torch.amp.autocast(device, dtype=torch.bfloat16)` (:758), next to its fp32 run on the same inputs: greedy ids of both,
    and the teacher-forced latents (UnifiedVoice.forward, model_v2.py:596-646) of both on the fp32 ids.  This is what the
    benchmarked bf16 engine mode is gated against (tests/test_gpu_gpt.py::test_bf16_*)."""
    seed, B, L, lens, max_gen = 41, 3, 16, [16, 16, 16], 24      # no left padding: the embeddings stay bf16 as in the B=1 pipeline
    cfg = G.GPTConfig(layers=6, model_dim=256, heads=4, max_text_tokens=40, max_mel_tokens=60, number_text_tokens=200)
    sd = G.synth_weights(cfg, seed=seed)
    sd["mel_head.bias"][cfg.stop_mel_token] -= 5.0             # fixed-length decode: every row emits max_gen tokens
    g = torch.Generator().manual_seed(seed + 100)
    text = ragged_text(g, B, L, cfg.number_text_tokens, lens)
    style = torch.randn(1, 192, generator=g)
    emo_vec = torch.randn(1, cfg.model_dim, generator=g) * 0.1
    langs = torch.randint(0, cfg.n_langs, (B,), generator=g)
    gk = dict(do_sample=False, num_beams=1, repetition_penalty=10.0)
    out = {}
    # CUDA autocast runs layer_norm in fp32 (inputs and parameters cast up, fp32 result); the CPU autocast policy list lacks
    # it and the CPU kernel rejects bf16 parameters with an fp32 input.  Give the CPU run the CUDA policy.
    _ln = F.layer_norm

    def ln_fp32(x, shape, weight=None, bias=None, eps=1e-5):
        if x.dtype == torch.bfloat16 or (weight is not None and weight.dtype == torch.bfloat16):
            return _ln(x.float(), shape, None if weight is None else weight.float(), None if bias is None else bias.float(), eps)
        return _ln(x, shape, weight, bias, eps)

    F.layer_norm = ln_fp32
    try:
        _make_bf16_body(seed, B, L, lens, max_gen, cfg, sd, text, style, emo_vec, langs, gk, out)
    finally:
        F.layer_norm = _ln


def _make_bf16_body(seed, B, L, lens, max_gen, cfg, sd, text, style, emo_vec, langs, gk, out):
    for kv in (True, False):
        uv = build_reference(sd, cfg, kv_cache=kv)
        with torch.no_grad():
            c32, _ = uv.inference_speech(torch.zeros(1, 4, 2), text, langs=langs, emo_vec=emo_vec, campplus_embedding=style,
                                         max_generate_length=max_gen, **gk)
        uvb = build_reference(sd, cfg, kv_cache=kv).bfloat16()
        with torch.no_grad(), torch.amp.autocast("cpu", enabled=True, dtype=torch.bfloat16):
            # emo_vec comes out of merge_emovec under the same autocast in the pipeline (:759-765), i.e. in bf16
            c16, _ = uvb.inference_speech(torch.zeros(1, 4, 2), text, langs=langs, emo_vec=emo_vec.bfloat16(),
                                          campplus_embedding=style, max_generate_length=max_gen, **gk)
        out["codes_f32_kv" if kv else "codes_f32_nokv"] = c32.numpy()
        out["codes_bf16_kv" if kv else "codes_bf16_nokv"] = c16.numpy()
        agree = [int((c32[b, : min(c32.shape[1], c16.shape[1])] != c16[b, : min(c32.shape[1], c16.shape[1])]).nonzero()[0])
                 if (c32[b, : min(c32.shape[1], c16.shape[1])] != c16[b, : min(c32.shape[1], c16.shape[1])]).any() else -1
                 for b in range(B)]
        print(f"bf16 (kv_cache={kv}): reference fp32 ids {tuple(c32.shape)}, reference bf16 ids {tuple(c16.shape)}, "
              f"first divergence per row {agree} (-1 = none)")
        if kv:
            codes, uv32, uv16 = c32, uv, uvb
    tl = torch.tensor(lens)
    ml = torch.full((B,), codes.shape[1])
    conds = G.conds_latent_campplus(sd, style, emo_vec)
    spk = F.linear(style, sd["spk_emb_proj.weight"], sd["spk_emb_proj.bias"]).unsqueeze(0).repeat(B, 1, 1)
    with torch.no_grad():
        lat32 = uv32.forward(spk, text.clone(), tl, codes.clone(), ml, None, emo_vec=emo_vec.repeat(B, 1), do_spk_cond=False)
        with torch.amp.autocast("cpu", enabled=True, dtype=torch.bfloat16):
            lat16 = uv16.forward(spk.bfloat16(), text.clone(), tl, codes.clone(), ml, None, emo_vec=emo_vec.repeat(B, 1).bfloat16(),
                                 do_spk_cond=False)
        lat_o32 = G.forward_latent(sd, cfg, conds.repeat(B, 1, 1), text, tl, codes, ml)
        with G.numerics("bf16"):
            lat_o16 = G.forward_latent(G.bf16_weights(sd), cfg, conds.repeat(B, 1, 1), text, tl, codes, ml)
    lat16 = lat16.float()
    e = lambda a, b: float((a - b).abs().max())
    print(f"  latents {tuple(lat32.shape)} dtype of the reference bf16 run: {lat16.dtype}; max|d| vs reference fp32: "
          f"reference-bf16 {e(lat16, lat32):.4f}, oracle bf16-contract {e(lat_o16, lat32):.4f}, oracle fp32 {e(lat_o32, lat32):.2e}; "
          f"oracle-bf16 vs reference-bf16 {e(lat_o16, lat16):.4f}")
    np.savez_compressed(os.path.join(GOLD, "gpt_bf16.npz"), text=text.numpy(), text_lens=tl.numpy(), style=style.numpy(),
                        emo_vec=emo_vec.numpy(), langs=langs.numpy(), mel_codes=codes.numpy(), mel_lens=ml.numpy(),
                        latent_f32=lat32.numpy().astype(np.float32), latent_bf16=lat16.numpy().astype(np.float32),
                        seed=np.int64(seed), eos_bias=np.float64(-5.0), max_gen=np.int64(max_gen),
                        cfg=np.array([cfg.layers, cfg.model_dim, cfg.heads, cfg.max_text_tokens, cfg.max_mel_tokens,
                                      cfg.number_text_tokens]), **out)


if __name__ == "__main__":
    main()
