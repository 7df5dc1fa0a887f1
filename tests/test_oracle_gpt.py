"""Pin the GPT CPU oracle against fixtures minted by running the reference's own code (tools/make_golden_gpt.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import gpt_oracle as G


def load_case(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, f"gpt_{tag}.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]),
                      max_mel_tokens=int(c[4]), number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] += float(z["eos_bias"])
    return z, cfg, sd


def gen_params(z):
    g = z["gen"]
    return G.GenParams(do_sample=bool(g[0]), num_beams=int(g[1]), top_p=float(g[2]), top_k=int(g[3]),
                       temperature=float(g[4]), repetition_penalty=float(g[5]), length_penalty=float(g[6]),
                       typical_sampling=bool(len(g) > 7 and int(g[7])), typical_mass=float(g[8]) if len(g) > 8 else 0.9,
                       max_generate_length=int(z["max_gen"]))


@pytest.mark.parametrize("tag", ["greedy", "greedy_nokv", "sample", "beam", "beam_sample", "greedy_mid",
                                 "typical_sample", "typical_greedy", "typical_beam_sample"])
def test_codes_match_reference(golden_dir, tag):
    z, cfg, sd = load_case(golden_dir, tag)
    gp = gen_params(z)
    u = torch.from_numpy(z["uniforms"])
    if gp.num_beams == 1:
        u = u[..., 0]
    conds = G.conds_latent_campplus(sd, torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"]))
    with torch.no_grad():
        codes = G.inference_speech(sd, cfg, conds, torch.from_numpy(z["text"]), torch.from_numpy(z["langs"]), gp,
                                   uniforms=u, kv_cache=bool(z["kv_cache"]))
    assert codes.shape == z["codes"].shape
    assert np.array_equal(codes.numpy(), z["codes"])


def test_latent_pass_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "gpt_latent.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]),
                      max_mel_tokens=int(c[4]), number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] += float(z["eos_bias"])
    B = z["text"].shape[0]
    conds = G.conds_latent_campplus(sd, torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"])).repeat(B, 1, 1)
    with torch.no_grad():
        lat = G.forward_latent(sd, cfg, conds, torch.from_numpy(z["text"]), torch.from_numpy(z["text_lens"]),
                               torch.from_numpy(z["mel_codes"]), torch.from_numpy(z["mel_lens"]))
    np.testing.assert_allclose(lat.numpy(), z["latent"], rtol=0, atol=2e-5)


def test_padding_invariance(golden_dir):
    """tests/padding_test.py:45-99 of the reference, as a property: unpadded == left-padded == batched."""
    z, cfg, sd = load_case(golden_dir, "greedy")
    gp = gen_params(z)
    text = torch.from_numpy(z["text"])
    langs = torch.from_numpy(z["langs"])
    conds = G.conds_latent_campplus(sd, torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"]))
    for b in range(text.shape[0]):
        n = int((text[b] != cfg.stop_text_token).sum())
        with torch.no_grad():
            solo = G.inference_speech(sd, cfg, conds, text[b:b + 1, :n], langs[b:b + 1], gp)
        ref = z["codes"][b]
        k = min(solo.shape[1], len(ref))
        assert np.array_equal(solo[0, :k].numpy(), ref[:k])
        assert np.all(ref[k:] == cfg.stop_mel_token)


def test_kv_cache_position_quirk(golden_dir):
    """kv_cache=True feeds mel position k+1 to the k-th generated token; kv_cache=False re-embeds 0..n-1
     -- the two modes must differ on the same inputs."""
    z, cfg, sd = load_case(golden_dir, "greedy_mid")
    gp = gen_params(z)
    conds = G.conds_latent_campplus(sd, torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"]))
    args = (sd, cfg, conds, torch.from_numpy(z["text"])[:1], torch.from_numpy(z["langs"])[:1], gp)
    with torch.no_grad():
        a = G.inference_speech(*args, kv_cache=True)
        b = G.inference_speech(*args, kv_cache=False)
    assert a[0, 0] == b[0, 0]                       # first token: both use position 0
    assert not torch.equal(a[:, :6], b[:, :6])


def test_processors_vs_installed_hf():
    tr = pytest.importorskip("transformers")
    from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
                                                        TopKLogitsWarper, TopPLogitsWarper)
    g = torch.Generator().manual_seed(0)
    scores = torch.randn(4, 500, generator=g) * 3
    ids = torch.randint(0, 500, (4, 17), generator=g)
    for min_keep, nb in ((1, 1), (2, 3)):
        ref = RepetitionPenaltyLogitsProcessor(10.0)(ids, scores.clone())
        ref = TemperatureLogitsWarper(0.8)(ids, ref)
        ref = TopKLogitsWarper(30, min_tokens_to_keep=min_keep)(ids, ref)
        ref = TopPLogitsWarper(0.8, min_tokens_to_keep=min_keep)(ids, ref)
        mine = G.process_scores(scores.clone(), ids, G.GenParams(do_sample=True, num_beams=nb))
        assert torch.equal(torch.isinf(ref), torch.isinf(mine))
        assert torch.allclose(ref[~torch.isinf(ref)], mine[~torch.isinf(mine)], atol=0, rtol=0)


def test_v1_codes_and_latent_match_reference(golden_dir):
    """IndexTTS-1/1.5 classes (indextts/gpt/model.py): greedy kv_cache=False decode + return_latent pass, fixture minted by
    running the reference's own v1 inference_speech / forward (tools/make_golden_gpt.py make_v1)."""
    z = np.load(os.path.join(golden_dir, "gpt_v1.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]),
                      max_mel_tokens=int(c[4]), number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] += float(z["eos_bias"])
    conds, text = torch.from_numpy(z["conds"]), torch.from_numpy(z["text"])
    gp = G.GenParams(do_sample=False, num_beams=1, repetition_penalty=10.0, max_generate_length=int(z["max_gen"]))
    with torch.no_grad():
        codes = G.inference_speech(sd, cfg, conds, text, None, gp, kv_cache=False)
        lat = G.forward_latent_v1(sd, cfg, conds.repeat(text.shape[0], 1, 1), text, torch.from_numpy(z["text_lens"]),
                                  torch.from_numpy(z["mel_codes"]), torch.from_numpy(z["code_lens"]) * 1024)
    assert np.array_equal(codes.numpy(), z["codes"])
    np.testing.assert_allclose(lat.numpy(), z["latent"], rtol=0, atol=5e-6)


def test_bf16_contract_vs_reference_bf16_fixture(golden_dir):
    """tests/golden/gpt_bf16.npz = the reference's own classes run in fp32 and in its bf16 mode (.bfloat16() + autocast,
    indextts/infer_v2_5.py:143-146,758) on the same inputs.  The oracle's fp32 mode reproduces the fp32 run; its "bf16" mode
    (the HIP engine's contract: bf16 GEMM operands and K/V, fp32 everything else) must stay at least as close to the fp32
    reference as the reference's own bf16 arithmetic does -- latents and teacher-forced logits."""
    import torch.nn.functional as F
    z = np.load(os.path.join(golden_dir, "gpt_bf16.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]),
                      max_mel_tokens=int(c[4]), number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] += float(z["eos_bias"])
    text, tl = torch.from_numpy(z["text"]), torch.from_numpy(z["text_lens"])
    codes, ml = torch.from_numpy(z["mel_codes"]), torch.from_numpy(z["mel_lens"])
    style, emo, langs = torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"]), torch.from_numpy(z["langs"])
    conds = G.conds_latent_campplus(sd, style, emo)
    B = text.shape[0]
    lat32, lat_ref16 = torch.from_numpy(z["latent_f32"]), torch.from_numpy(z["latent_bf16"])
    with torch.no_grad():
        lo32 = G.forward_latent(sd, cfg, conds.repeat(B, 1, 1), text, tl, codes, ml)
        with G.numerics("bf16"):
            lo16 = G.forward_latent(G.bf16_weights(sd), cfg, conds.repeat(B, 1, 1), text, tl, codes, ml)
        for kv in (True, False):
            ids = G.inference_speech(sd, cfg, conds, text, langs, G.GenParams(max_generate_length=int(z["max_gen"])), kv_cache=kv)
            assert np.array_equal(ids.numpy(), z["codes_f32_kv" if kv else "codes_f32_nokv"])
    assert float((lo32 - lat32).abs().max()) < 5e-5
    e16, eref = float((lo16 - lat32).abs().max()), float((lat_ref16 - lat32).abs().max())
    W, b = sd["mel_head.weight"], sd["mel_head.bias"]
    lg32 = F.linear(lat32, W, b)
    d16 = float((F.linear(lo16.bfloat16().float(), W.bfloat16().float(), b) - lg32).abs().max())
    dref = float((F.linear(lat_ref16.bfloat16(), W.bfloat16(), b.bfloat16()).float() - lg32).abs().max())
    assert e16 <= 0.03 and d16 <= 0.05                            # the bounds tests/test_gpu_gpt.py holds the engine to
    assert e16 <= eref and d16 <= dref
    assert G._NUMERICS == "f32"                                    # the context manager restored the default
