"""GPU parity tests: HIP GPT decoder (through the C ABI) vs the CPU oracle and the reference-minted goldens.

Bar (north_star): speech-token ids BIT-EXACT under greedy decode vs the reference CPU path (f32 engine mode).
Sampled modes consume the fixture's uniform stream and must reproduce the reference ids as well.
The bf16 engine mode (the benchmarked one) is reported against the f32 ids (agreement prefix), not gated bit-exact.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import gpt_oracle as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def load_case(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, f"gpt_{tag}.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]),
                      max_mel_tokens=int(c[4]), number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] += float(z["eos_bias"])
    return z, cfg, sd


def engine(cfg, sd, precision="fp32"):
    from indextts_amd import gpt
    m = gpt.UnifiedVoice(spk_cond_mode="campplus", layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads,
                         max_text_tokens=cfg.max_text_tokens, max_mel_tokens=cfg.max_mel_tokens,
                         number_text_tokens=cfg.number_text_tokens, precision=precision, device=DEV)
    m.load_state_dict(sd)
    return m


@pytest.mark.parametrize("prec,prefill", [(0, False), (0, True), (1, False), (1, True)])
@pytest.mark.parametrize("M,K,N", [(5, 128, 40), (64, 1280, 384), (70, 256, 8194), (130, 5120, 96)])
def test_gemm_vs_torch(prec, prefill, M, K, N):
    from indextts_amd import gpt
    g = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    if prec == 1:
        a_in = a.bfloat16()
        ref = a_in.float() @ w.bfloat16().float() + bias
        tol = 2e-3
    else:
        a_in = a
        ref = a.double() @ w.double() + bias.double()
        tol = 2e-5
    wp = gpt.pack_gemm_weight(w, prec).to(DEV)
    y = gpt.gemm(a_in.to(DEV), wp, bias.to(DEV), N, prec, prefill_tiles=prefill).cpu()
    assert (y.double() - ref.double()).abs().max() < tol * max(1.0, float(ref.abs().max()))


def test_layernorm_vs_torch():
    from indextts_amd import gpt
    g = torch.Generator().manual_seed(2)
    for D in (128, 256, 1280):
        x = torch.randn(9, D, generator=g) * 3 + 1
        g1, b1, g2, b2 = (torch.randn(D, generator=g) for _ in range(4))
        ref = torch.nn.functional.layer_norm(x, (D,), g1, b1, 1e-5)
        y = gpt.layernorm(x.to(DEV), g1.to(DEV), b1.to(DEV)).cpu()
        assert (y - ref).abs().max() < 2e-5
        ref2 = torch.nn.functional.layer_norm(ref, (D,), g2, b2, 1e-5)
        y2 = gpt.layernorm(x.to(DEV), g1.to(DEV), b1.to(DEV), g2.to(DEV), b2.to(DEV)).cpu()
        assert (y2 - ref2).abs().max() < 5e-5


@pytest.mark.parametrize("M,K,N", [(1, 1280, 3840), (3, 1280, 5120), (4, 1280, 8194), (2, 256, 96), (4, 512, 1536),
                                   (5, 1280, 3840), (8, 1280, 5120), (11, 1280, 3840), (16, 1280, 5120), (13, 256, 96), (16, 512, 1536)])
@pytest.mark.parametrize("pending", [False, True])
def test_layernorm_fused_decode_gemm_is_bitwise_the_two_launches(M, K, N, pending):
    """gemm_decode_ln_kernel / gemm_decode_lnw_kernel (decode steps of 1-16 rows -- 4 waves up to 4 rows, the wide kernel above: LayerNorm computed inside the consuming GEMM's operand staging, with the split-K reduce of
    the previous GEMM's partials + bias + residual) against the two launches it replaces -- itts_layernorm_forward -> bf16 -> itts_gemm_forward
    (the 16-row slab kernel): output BITWISE equal, and so is the updated residual row it writes back."""
    from indextts_amd import gpt
    g = torch.Generator().manual_seed(7 * M + K + N + int(pending))
    x = (torch.randn(M, K, generator=g) * 2 + 0.3).to(DEV)
    g1, b1 = (1 + 0.1 * torch.randn(K, generator=g)).to(DEV), (0.1 * torch.randn(K, generator=g)).to(DEV)
    w = torch.randn(K, N, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g).to(DEV)
    wp = gpt.pack_gemm_weight(w, 1).to(DEV)
    partial = bias_prev = None
    xr = x
    if pending:
        partial = (torch.randn(4, M, K, generator=g) * 0.5).to(DEV)
        bias_prev = (0.1 * torch.randn(K, generator=g)).to(DEV)
        xr = (x + ((partial[0] + partial[1]) + (partial[2] + partial[3]))) + bias_prev          # ln_row's order of additions (exact IEEE adds)
    from indextts_amd import _lib
    ref = gpt.gemm(gpt.layernorm(xr, g1, b1).bfloat16(), wp, bias, N, 1)
    # 5-16 rows: the wide kernel with 2 / 4 n-tiles per block (weights on waves 0-3, LayerNorm on waves 4-7)
    for nt in ((2, 4) if M > 4 else (2,)):
        with _lib.option_scope(decode_ln_nt=nt):
            out, x_out = gpt.gemm_ln(x, g1, b1, wp, bias, N, partial=partial, bias_prev=bias_prev)
        assert torch.equal(out, ref), (nt, float((out - ref).abs().max()))
        if pending:
            assert torch.equal(x_out, xr), nt
    assert float(out.abs().mean()) > 1e-2


def test_fused_layernorm_decode_steps_equal_unfused(tmp_path):
    """Whole decode loops of 1 .. 17 rows (greedy, sampled, 3-beam beam-sample of one utterance) with the LayerNorm-fused GEMMs (default 1: steps of
    1-8 rows; 2: up to 16 rows) and without (option decode_fuse_ln = 0): identical ids -- at the
    production width (K = 1280: NV = 5) and at 256."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuse_ln_probe.py")
    for big in ("1", "0"):
        outs = []
        for v in ("0", "1", "2", "2,decode_ln_nt=4"):       # (1: fused up to 8 rows; 2: up to 16, wide kernel with 2 / 4 n-tiles per block)
            env = dict(os.environ, PROBE_OPTS=f"decode_fuse_ln={v}", PROBE_BIG=big)
            r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1])
        assert len(set(outs)) == 1, outs


def test_topk_bisection_equals_radix_select():
    """The top-k threshold of the sampling / beam kernels by ballot bisection (option sample_radix = 0) and by the 4-pass radix select (= 1): the
    same k-th key, hence identical ids over sampled and 3-beam beam-sample decode loops (both paths are also gated against the oracle's and the
    reference's ids by the golden tests, which run the default)."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuse_ln_probe.py")
    outs = []
    for v in ("0", "1"):
        env = dict(os.environ, PROBE_OPTS=f"sample_radix={v}", PROBE_BIG="0")
        r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1])
    assert len(set(outs)) == 1, outs


def test_attention_geometries_agree():
    """The KV-cache attention kernel maps its 16 canonical key streams onto 4, 8 or 16 waves per block (picked by the launch's block count; option
    attn_waves forces one): every geometry performs the same operations in the same order -> identical ids over greedy / sampled / beam loops of
    1-17 rows, bf16, at both probe widths (the f32 engine's geometries are compared in test_attention_geometries_agree_f32)."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuse_ln_probe.py")
    for big in ("1", "0"):
        outs = []
        for v in ("4", "8", "16"):
            env = dict(os.environ, PROBE_OPTS=f"attn_waves={v}", PROBE_BIG=big)
            r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1])
        assert outs[0] == outs[1] == outs[2], outs


def test_attention_geometries_agree_f32(golden_dir):
    """f32 engine, one process: the teacher-forced latents (prefill attention, one block per query) and 30 greedy steps of a ragged batch come out
    BITWISE equal with 4, 8 and 16 waves per attention block (option attn_waves; changing it also retires the cached decode graph)."""
    from indextts_amd import _lib
    z = np.load(os.path.join(golden_dir, "gpt_latent.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]),
                      max_mel_tokens=int(c[4]), number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] -= 1e4
    m = engine(cfg, sd, "fp32")
    B = z["text"].shape[0]
    style, emo = torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"])
    conds, _ = m.conds_latent(style, emo)
    text = torch.from_numpy(z["text"])
    outs = []
    for nw in (4, 8, 16, 0):
        with _lib.option_scope(attn_waves=nw):
            lat = m.forward_latent(conds.repeat(B, 1, 1), text, torch.from_numpy(z["text_lens"]),
                                   torch.from_numpy(z["mel_codes"]), torch.from_numpy(z["mel_lens"])).cpu()
            ids, _ = m.inference_speech(None, text, langs=torch.full((B,), 1), emo_vec=emo, campplus_embedding=style, max_generate_length=30,
                                        do_sample=False, num_beams=1, repetition_penalty=10.0)
        outs.append((lat, ids.cpu()))
    assert _lib.get_option("attn_waves") == 0
    for lat, ids in outs[1:]:
        assert torch.equal(lat, outs[0][0]) and torch.equal(ids, outs[0][1])
    assert float(outs[0][0].abs().mean()) > 1e-3


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_prefill_mfma_attention_vs_canonical_streams(golden_dir, prec):
    """attn_prefill_mfma_kernel (causal flash attention of S > 1 passes on the matrix pipe: v_mfma_f32_16x16x4_f32 in the f32 engine, bf16 MFMAs with
    hi + lo query / probability operands in the bf16 engine) against attn_kernel (one block per query, canonical key streams; option prefill_attn = 0):
    teacher-forced latents of the reference-minted fixture (f32: to accumulation-order noise; bf16: inside one bf16 step of the hidden state), and, f32,
    the greedy ids of a RAGGED batch (left-padded prefill, pad[b] > 0) with the KV cache and without it -- then every step is an S > 1 pass."""
    from indextts_amd import _lib
    z = np.load(os.path.join(golden_dir, "gpt_latent.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]),
                      max_mel_tokens=int(c[4]), number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] -= 1e4
    m = engine(cfg, sd, prec)
    B = z["text"].shape[0]
    style, emo = torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"])
    conds, _ = m.conds_latent(style, emo)
    text, tl = torch.from_numpy(z["text"]), torch.from_numpy(z["text_lens"])
    assert int(tl.min()) < int(tl.max())                        # ragged: the prefill is left-padded
    outs = {}
    for pa in (0, 1):
        with _lib.option_scope(prefill_attn=pa):
            lat = m.forward_latent(conds.repeat(B, 1, 1), text, tl, torch.from_numpy(z["mel_codes"]), torch.from_numpy(z["mel_lens"])).cpu()
            ids = {}
            for kv in (True, False):
                m.post_init_gpt2_config(kv_cache=kv)
                ids[kv], _ = m.inference_speech(None, text, langs=torch.full((B,), 1), emo_vec=emo, campplus_embedding=style, max_generate_length=24,
                                                do_sample=False, num_beams=1, repetition_penalty=10.0)
            m.post_init_gpt2_config(kv_cache=True)
        outs[pa] = (lat, {k: v.cpu() for k, v in ids.items()})
    lat0, lat1 = outs[0][0], outs[1][0]
    rel = float((lat1 - lat0).pow(2).mean().sqrt() / lat0.pow(2).mean().sqrt())
    print(f"prefill MFMA attention vs canonical streams ({prec}): latent rms difference {rel:.3e} of the latent rms")
    assert torch.isfinite(lat1).all() and float(lat0.abs().mean()) > 1e-3
    assert rel < (2e-6 if prec == "fp32" else 1.5e-2), rel
    if prec == "fp32":                                          # (cached and uncached decoding run different GEMM kernels: each is compared with itself)
        for kv in (True, False):
            assert torch.equal(outs[1][1][kv], outs[0][1][kv]), kv


def run_case(m, z, cfg, sd):
    g = z["gen"]
    kw = dict(do_sample=bool(g[0]), num_beams=int(g[1]), top_p=float(g[2]), top_k=int(g[3]), temperature=float(g[4]),
              repetition_penalty=float(g[5]), length_penalty=float(g[6]))
    if len(g) > 7 and int(g[7]):
        kw.update(typical_sampling=True, typical_mass=float(g[8]))
    u = torch.from_numpy(z["uniforms"])
    if kw["num_beams"] == 1:
        u = u[..., 0]
    m.post_init_gpt2_config(kv_cache=bool(z["kv_cache"]))
    codes, _ = m.inference_speech(None, torch.from_numpy(z["text"]), langs=torch.from_numpy(z["langs"]),
                                  emo_vec=torch.from_numpy(z["emo_vec"]), campplus_embedding=torch.from_numpy(z["style"]),
                                  max_generate_length=int(z["max_gen"]), uniforms=u if kw["do_sample"] else None, **kw)
    return codes.cpu().numpy()


@pytest.mark.parametrize("tag", ["greedy", "greedy_mid", "greedy_nokv", "sample", "typical_sample", "typical_greedy"])
@pytest.mark.parametrize("use_graph", [True, False])
def test_codes_bit_exact_vs_reference_golden(golden_dir, tag, use_graph):
    """f32 engine mode: ids identical to the ids the REFERENCE's own generate() produced (ragged left-padded batch,
    EOS at ragged steps, repetition penalty 10, kv-cache position quirk / no-kv rule, top-k/top-p sampling)."""
    z, cfg, sd = load_case(golden_dir, tag)
    m = engine(cfg, sd, "fp32")
    m.use_graph = use_graph
    codes = run_case(m, z, cfg, sd)
    assert codes.shape == z["codes"].shape, (codes.shape, z["codes"].shape)
    if not np.array_equal(codes, z["codes"]):
        bad = np.argwhere(codes != z["codes"])
        pytest.fail(f"first divergence at (row, step) = {bad[0].tolist()}: got {codes[tuple(bad[0])]} want {z['codes'][tuple(bad[0])]}")


@pytest.mark.parametrize("tag", ["beam", "beam_sample", "typical_beam_sample"])
@pytest.mark.parametrize("use_graph", [True, False])
def test_beam_codes_bit_exact_vs_reference_golden(golden_dir, tag, use_graph):
    """Beam search and 3-beam beam-sample (the reference DEFAULT mode, infer_v2_5.py:732-740): device beam step +
    scorer + KV row map + host finalize reproduce the ids of the reference's own _beam_search/BeamSearchScorer."""
    z, cfg, sd = load_case(golden_dir, tag)
    m = engine(cfg, sd, "fp32")
    m.use_graph = use_graph
    codes = run_case(m, z, cfg, sd)
    assert codes.shape == z["codes"].shape, (codes.shape, z["codes"].shape)
    if not np.array_equal(codes, z["codes"]):
        bad = np.argwhere(codes != z["codes"])
        pytest.fail(f"first divergence at (row, step) = {bad[0].tolist()}: got {codes[tuple(bad[0])]} want {z['codes'][tuple(bad[0])]}")


def test_padding_invariance_on_device(golden_dir):
    """tests/padding_test.py of the reference as a device property: a row decoded alone == the row inside the batch."""
    z, cfg, sd = load_case(golden_dir, "greedy_mid")
    m = engine(cfg, sd, "fp32")
    text = torch.from_numpy(z["text"])
    for b in range(text.shape[0]):
        n = int((text[b] != cfg.stop_text_token).sum())
        codes, _ = m.inference_speech(None, text[b:b + 1, :n], langs=torch.from_numpy(z["langs"])[b:b + 1],
                                      emo_vec=torch.from_numpy(z["emo_vec"]), campplus_embedding=torch.from_numpy(z["style"]),
                                      max_generate_length=int(z["max_gen"]), do_sample=False, repetition_penalty=10.0)
        ref = z["codes"][b]
        k = min(codes.shape[1], len(ref))
        assert np.array_equal(codes[0, :k].cpu().numpy(), ref[:k])


def test_bf16_mode_tracks_f32(golden_dir):
    """bf16 weights/KV: the first token must agree and the agreement prefix is reported (not gated bit-exact)."""
    z, cfg, sd = load_case(golden_dir, "greedy_mid")
    m = engine(cfg, sd, "bf16")
    codes = run_case(m, z, cfg, sd)
    ref = z["codes"]
    n = min(codes.shape[1], ref.shape[1])
    agree = [int((np.cumprod(codes[b, :n] == ref[b, :n])).sum()) for b in range(ref.shape[0])]
    print("bf16 agreement prefix per row:", agree, "of", n)
    assert all(a >= 1 for a in agree)


def test_latent_pass_vs_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "gpt_latent.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]),
                      max_mel_tokens=int(c[4]), number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] += float(z["eos_bias"])
    m = engine(cfg, sd, "fp32")
    B = z["text"].shape[0]
    conds, _ = m.conds_latent(torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"]))
    lat = m.forward_latent(conds.repeat(B, 1, 1), torch.from_numpy(z["text"]), torch.from_numpy(z["text_lens"]),
                           torch.from_numpy(z["mel_codes"]), torch.from_numpy(z["mel_lens"])).cpu().numpy()
    assert lat.shape == z["latent"].shape
    np.testing.assert_allclose(lat, z["latent"], rtol=0, atol=5e-5)


def test_v1_decode_and_latent_vs_reference_golden(golden_dir):
    """IndexTTS-1/1.5 mirror (`UnifiedVoiceV1`, reference indextts/gpt/model.py): BASELINE configs[0] in miniature --
    greedy, kv_cache=False position rule, 32-token conditioning latent, then `gpt(..., return_latent=True)`."""
    from indextts_amd import gpt
    z = np.load(os.path.join(golden_dir, "gpt_v1.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]),
                      max_mel_tokens=int(c[4]), number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] += float(z["eos_bias"])
    m = gpt.UnifiedVoiceV1(layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, max_text_tokens=cfg.max_text_tokens,
                           max_mel_tokens=cfg.max_mel_tokens, number_text_tokens=cfg.number_text_tokens,
                           precision="fp32", device=DEV)
    m.load_state_dict(sd)
    m.post_init_gpt2_config(kv_cache=False)
    conds, text = torch.from_numpy(z["conds"]), torch.from_numpy(z["text"])
    m.conditioning_fn = lambda mel, lengths=None: conds.to(DEV)          # stands in for Conformer + Perceiver
    codes = m.inference_speech(torch.zeros(1, 100, 7), text, cond_mel_lengths=torch.tensor([7]),
                               max_generate_length=int(z["max_gen"]), do_sample=False, num_beams=1, repetition_penalty=10.0)
    assert np.array_equal(codes.cpu().numpy(), z["codes"])
    B = text.shape[0]
    lat = m(conds.repeat(B, 1, 1), text, torch.from_numpy(z["text_lens"]), torch.from_numpy(z["mel_codes"]),
            torch.from_numpy(z["code_lens"]) * m.mel_length_compression, cond_mel_lengths=torch.tensor([7]),
            return_latent=True, clip_inputs=False, conds_latent=conds.repeat(B, 1, 1))
    assert lat.shape == z["latent"].shape
    np.testing.assert_allclose(lat.cpu().numpy(), z["latent"], rtol=0, atol=5e-5)
    with pytest.raises(NotImplementedError):
        m.conditioning_fn = None
        m.inference_speech(torch.zeros(1, 100, 7), text)


def test_prefill_gemm_kernels_agree():
    """bf16 prefill: the LDS-DMA 128x128 tile kernel and the direct-load kernel issue the same MFMA on the same fragments
    in the same K order, so the teacher-forced latents and the decoded ids must be BITWISE equal between the two."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prefill_probe.py")
    outs = []
    for v in ("0", "1"):
        env = dict(os.environ, PROBE_OPTS=f"prefill_gemm={v}")
        r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1]
        outs.append(line)
    assert float(outs[0].split()[2]) > 1e-3          # the latents are not trivially zero
    assert outs[0] == outs[1], outs


def test_prefill_tile_kernels_agree():
    """GPT prefill / latent pass (EPI_QKV cache append, GELU, residual, plain store) through the 256 x 256 and 256 x 128 tile
    kernels (option tile256 = 1 / 2 forces them for every shape) vs the 128 x 128 kernel: bitwise equal latents and ids."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prefill_probe.py")
    outs = []
    for v in ("0", "1", "2"):
        env = dict(os.environ, PROBE_OPTS=f"tile256={v}", PROBE_BIG="1")
        r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1])
    assert outs[0] == outs[1] == outs[2], outs


def test_decode_gemm_kernels_agree():
    """bf16 decode: the LDS-DMA slab kernel keeps the register-path kernels' k-block split and reduction order -> BITWISE
    equal latents/ids (full-width stack: K slices of 1280, split-K of the 5120-deep projection; and the 16-row slab)."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prefill_probe.py")
    for big in ("1", "0"):                 # 40 rows of the full-width stack (64-row slab), 5 rows of a small one (16-row slab)
        outs = []
        for v in ("0", "1"):
            env = dict(os.environ, PROBE_OPTS=f"decode_gemm={v}", PROBE_BIG=big)
            r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1])
        assert outs[0] == outs[1], outs


@pytest.mark.parametrize("dim,rows", [(128, 5), (384, 40), (640, 20)])
def test_decode_gemm_odd_kblock_slices(dim, rows):
    """bf16 decode at widths whose 4-way split-K slices hold an ODD number of 32-wide k-blocks (D/128 odd): the slab kernel's
    128-byte k-pair DMA cannot represent such a slice, so those GEMMs must take the register path -- outputs bitwise equal
    to a run with the slab kernel disabled (silent wrong partials at D = 128, 384, 640, ...)."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prefill_probe.py")
    outs = []
    for v in ("0", "1"):
        env = dict(os.environ, PROBE_OPTS=f"decode_gemm={v}", PROBE_DIM=str(dim), PROBE_B=str(rows))
        r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1])
    assert outs[0] == outs[1], outs


def test_typical_sampling_at_production_vocab():
    """TypicalLogitsWarper on the device loop at V = 8194 (two score rows = 65.5 KB of dynamic LDS, above the default
    64 KiB): ids equal the CPU oracle in sample and beam-sample modes."""
    cfg = G.GPTConfig(layers=2, model_dim=128, heads=2, max_text_tokens=20, max_mel_tokens=40, number_text_tokens=60)
    assert cfg.number_mel_codes == 8194
    sd = G.synth_weights(cfg, seed=21)
    g = torch.Generator().manual_seed(4)
    text = torch.randint(2, 60, (2, 7), generator=g)
    style = torch.randn(1, 192, generator=g)
    emo = torch.randn(1, cfg.model_dim, generator=g) * 0.1
    langs = torch.tensor([1, 2])
    conds = G.conds_latent_campplus(sd, style, emo)
    m = engine(cfg, sd, "fp32")
    for nb in (1, 2):
        n = 6
        u = torch.rand(n, 2, generator=g, dtype=torch.float64) if nb == 1 else torch.rand(n, 2, 2 * nb, generator=g, dtype=torch.float64)
        gp = G.GenParams(do_sample=True, num_beams=nb, top_k=30, top_p=0.8, temperature=0.8, repetition_penalty=10.0,
                         max_generate_length=n, typical_sampling=True, typical_mass=0.9)
        with torch.no_grad():
            ref = G.inference_speech(sd, cfg, conds, text, langs, gp, uniforms=u)
        ids, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=n,
                                    do_sample=True, num_beams=nb, top_k=30, top_p=0.8, temperature=0.8, repetition_penalty=10.0,
                                    typical_sampling=True, typical_mass=0.9, uniforms=u)
        assert torch.equal(ids.cpu(), ref), (nb, ids.cpu().tolist(), ref.tolist())


def test_decode_graph_is_cached_across_calls():
    """The instantiated decode-step hipGraph lives in the engine handle: a second call with the same batch rows, prompt-length
    bucket (multiples of 32), max_new_tokens and generation parameters replays it; results stay bit-exact vs the oracle; a
    different seed does not force a new capture (the seed is read from device memory); another bucket does."""
    cfg = G.GPTConfig(layers=2, model_dim=128, heads=2, max_text_tokens=80, max_mel_tokens=40, number_text_tokens=60)
    sd = G.synth_weights(cfg, seed=11)
    m = engine(cfg, sd, "fp32")
    g = torch.Generator().manual_seed(1)
    style, emo = torch.randn(1, 192, generator=g), torch.randn(1, 128, generator=g) * 0.1
    conds = G.conds_latent_campplus(sd, style, emo)
    langs = torch.full((3,), 2)

    def run(n_text, **kw):
        text = torch.randint(2, 60, (3, n_text), generator=torch.Generator().manual_seed(n_text))
        ids, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=10,
                                    num_beams=1, repetition_penalty=10.0, **kw)
        return text, ids.cpu()

    s0 = m.graph_stats()
    text, ids = run(20, do_sample=False)                             # S = 3 + 22 + 1 = 26 -> bucket 32
    with torch.no_grad():
        ref = G.inference_speech(sd, cfg, conds, text, langs, G.GenParams(max_generate_length=10))
    assert torch.equal(ids, ref)
    s1 = m.graph_stats()
    assert (s1["captures"] - s0["captures"], s1["hits"] - s0["hits"]) == (1, 0)
    text2, ids2 = run(24, do_sample=False)                           # S = 30: same bucket, same graph
    with torch.no_grad():
        ref2 = G.inference_speech(sd, cfg, conds, text2, langs, G.GenParams(max_generate_length=10))
    assert torch.equal(ids2, ref2)
    s2 = m.graph_stats()
    assert (s2["captures"] - s1["captures"], s2["hits"] - s1["hits"]) == (0, 1)
    run(40, do_sample=False)                                         # S = 46: next bucket
    s3 = m.graph_stats()
    assert s3["captures"] - s2["captures"] == 1
    a = run(20, do_sample=True, top_k=30, top_p=0.8, temperature=1.5, seed=1)[1]
    b = run(20, do_sample=True, top_k=30, top_p=0.8, temperature=1.5, seed=2)[1]
    c = run(20, do_sample=True, top_k=30, top_p=0.8, temperature=1.5, seed=1)[1]
    s4 = m.graph_stats()
    assert s4["captures"] - s3["captures"] == 1 and s4["hits"] - s3["hits"] == 2
    assert torch.equal(a[:, : c.shape[1]], c[:, : a.shape[1]]) and not torch.equal(a, b)


def test_typical_mass_validation():
    from indextts_amd import gpt
    cfg = G.GPTConfig(layers=1, model_dim=128, heads=2, max_text_tokens=20, max_mel_tokens=30, number_text_tokens=50)
    m = engine(cfg, G.synth_weights(cfg, seed=3), "fp32")
    with pytest.raises(ValueError):
        m.inference_speech(None, torch.randint(2, 50, (1, 5)), emo_vec=torch.zeros(1, 128), campplus_embedding=torch.zeros(1, 192),
                           typical_sampling=True, typical_mass=1.5, max_generate_length=3)
    with pytest.raises(ValueError, match="max_text_tokens"):          # 21 text tokens + start/stop need 23 of the 22 positions
        m.inference_speech(None, torch.randint(2, 50, (1, 21)), emo_vec=torch.zeros(1, 128), campplus_embedding=torch.zeros(1, 192),
                           max_generate_length=3)


def test_full_size_greedy_vs_oracle():
    """IndexTTS-2.5 sized stack (24 x 1280, 20 heads), B=3 ragged, 10 steps: ids identical to the CPU oracle."""
    cfg = G.GPTConfig(max_text_tokens=120, max_mel_tokens=200)
    sd = G.synth_weights(cfg, seed=1234)
    g = torch.Generator().manual_seed(5)
    text = torch.randint(2, cfg.number_text_tokens, (3, 24), generator=g)
    text[1, 15:] = 1
    text[2, 9:] = 1
    style = torch.randn(1, 192, generator=g)
    emo = torch.randn(1, cfg.model_dim, generator=g) * 0.1
    langs = torch.tensor([3, 3, 7])
    gp = G.GenParams(max_generate_length=10)
    conds = G.conds_latent_campplus(sd, style, emo)
    trace = {}
    with torch.no_grad():
        ref = G.inference_speech(sd, cfg, conds, text, langs, gp, trace=trace).numpy()
    m = engine(cfg, sd, "fp32")
    codes, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style, max_generate_length=10,
                                  do_sample=False, repetition_penalty=10.0)
    codes = codes.cpu().numpy()
    top2 = [torch.topk(l, 2, dim=-1).values for l in trace["logits"]]
    margin = min(float((t[:, 0] - t[:, 1]).min()) for t in top2)
    print("min top-2 logit margin of the oracle run:", margin)
    assert np.array_equal(codes, ref), (codes, ref)


# ================================================================================================================
# The benchmarked mode (bf16) gated against the reference-minted bf16/fp32 fixture and, at full size, the CPU oracle.
#   Contract of the engine's bf16 mode (oracle.gpt_oracle.numerics("bf16")): GEMM weights, GEMM inputs and the K/V cache in
#   bf16; accumulation, residual stream, query, LayerNorm statistics, softmax and logits in fp32.
#   Bounds (written here, calibrated with the CPU restatement of that contract; the measured values are printed):
#     latents (final_norm output, O(1) entries): |engine - contract| <= 0.02 (bf16 rounding-boundary flips between two f32
#     summation orders; measured 8e-3) ; |engine - reference fp32| <= 0.03 (6 x 256; measured 0.013, the reference's own bf16: 0.029)
#     teacher-forced logits:   6 x 256 model  <= 0.05      24 x 1280 model  <= 0.25
#     greedy ids: identical to the fp32 ids at every step whose fp32 top-2 margin exceeds 2 x the logit bound; a row's
#     first divergence (if any) must sit on a step with a smaller margin (reported).
# ================================================================================================================
BF16_LATENT_VS_CONTRACT = 0.02
BF16_LATENT_VS_F32_SMALL = 0.03
BF16_LOGIT_BOUND_SMALL = 0.05
BF16_LOGIT_BOUND_FULL = 0.25


def _bf16_case(golden_dir):
    z = np.load(os.path.join(golden_dir, "gpt_bf16.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]),
                      max_mel_tokens=int(c[4]), number_text_tokens=int(c[5]))
    sd = G.synth_weights(cfg, seed=int(z["seed"]))
    sd["mel_head.bias"][cfg.stop_mel_token] += float(z["eos_bias"])
    return z, cfg, sd


def _rb(x):
    return x.bfloat16().float()


def test_bf16_latents_and_logits_vs_reference_fixture(golden_dir):
    """Teacher-forced pass of the bf16 engine on the ids the reference's fp32 run produced: against the CPU restatement of
    the engine's contract (tight), against the reference's fp32 latents/logits (stated bound), and next to the error the
    reference's OWN bf16 mode (.bfloat16() + autocast) makes on the same inputs."""
    z, cfg, sd = _bf16_case(golden_dir)
    m = engine(cfg, sd, "bf16")
    B = z["text"].shape[0]
    text, tl = torch.from_numpy(z["text"]), torch.from_numpy(z["text_lens"])
    codes, ml = torch.from_numpy(z["mel_codes"]), torch.from_numpy(z["mel_lens"])
    conds, _ = m.conds_latent(torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"]))
    lat = m.forward_latent(conds.repeat(B, 1, 1), text, tl, codes, ml).cpu()
    lat32, lat_ref16 = torch.from_numpy(z["latent_f32"]), torch.from_numpy(z["latent_bf16"])
    with torch.no_grad(), G.numerics("bf16"):
        lat_contract = G.forward_latent(G.bf16_weights(sd), cfg, conds.cpu().repeat(B, 1, 1), text, tl, codes, ml)
    e_contract = float((lat - lat_contract).abs().max())
    e_f32 = float((lat - lat32).abs().max())
    e_ref16 = float((lat_ref16 - lat32).abs().max())
    W, b = sd["mel_head.weight"], sd["mel_head.bias"]
    lg32 = F.linear(lat32, W, b)
    lg = F.linear(_rb(lat), _rb(W), b)                       # the engine's head GEMM: bf16 inputs, f32 accumulate
    lg_ref16 = F.linear(lat_ref16.bfloat16(), W.bfloat16(), b.bfloat16()).float()
    d_eng, d_ref16 = float((lg - lg32).abs().max()), float((lg_ref16 - lg32).abs().max())
    print(f"bf16 engine latents: vs contract {e_contract:.2e} (bound {BF16_LATENT_VS_CONTRACT}), vs reference fp32 {e_f32:.4f} "
          f"(bound {BF16_LATENT_VS_F32_SMALL}; the reference's own bf16 mode: {e_ref16:.4f}); logits vs fp32 {d_eng:.4f} "
          f"(bound {BF16_LOGIT_BOUND_SMALL}; reference bf16: {d_ref16:.4f})")
    assert e_contract <= BF16_LATENT_VS_CONTRACT
    assert e_f32 <= BF16_LATENT_VS_F32_SMALL
    assert d_eng <= BF16_LOGIT_BOUND_SMALL
    assert e_f32 <= 1.25 * e_ref16 and d_eng <= 1.25 * d_ref16      # no worse than the reference's own bf16 arithmetic


def _gated_agreement(ids, ref_ids, margins, bound, what):
    """ids/ref_ids (B, n); margins[step][row] = fp32 top-2 logit margin.  Returns the per-row first divergence (-1 = none)."""
    firsts = []
    for r in range(ref_ids.shape[0]):
        n = min(ids.shape[1], ref_ids.shape[1])
        ne = np.nonzero(ids[r, :n] != ref_ids[r, :n])[0]
        k = int(ne[0]) if len(ne) else -1
        firsts.append(k)
        if k >= 0:
            mk = float(margins[k][r])
            print(f"{what}: row {r} first divergence at step {k}, fp32 top-2 margin there {mk:.4f} (2 x bound = {2 * bound})")
            assert mk <= 2 * bound, f"{what}: row {r} diverged at step {k} where the fp32 margin {mk} exceeds 2 x {bound}"
    return firsts


@pytest.mark.parametrize("kv", [True, False])
def test_bf16_greedy_ids_gated_by_margin(golden_dir, kv):
    """Greedy decode in the benchmarked mode vs the ids of the reference's fp32 run (fixture): equal at every step whose
    fp32 top-2 margin exceeds twice the logit bound; the first divergence of a row (if any) is printed with its margin.
    The reference's own bf16 ids are in the fixture too: its first divergences are printed for comparison."""
    z, cfg, sd = _bf16_case(golden_dir)
    text, langs = torch.from_numpy(z["text"]), torch.from_numpy(z["langs"])
    style, emo = torch.from_numpy(z["style"]), torch.from_numpy(z["emo_vec"])
    ref = z["codes_f32_kv" if kv else "codes_f32_nokv"]
    trace = {}
    with torch.no_grad():
        oc = G.inference_speech(sd, cfg, G.conds_latent_campplus(sd, style, emo), text, langs,
                                G.GenParams(max_generate_length=int(z["max_gen"])), kv_cache=kv, trace=trace)
    assert np.array_equal(oc.numpy(), ref)                       # oracle fp32 == reference fp32 (pinning)
    # the margin that decides a greedy step is the one between the two best PROCESSED scores (repetition penalty applied)
    margins = [(lambda t: (t[:, 0] - t[:, 1]).numpy())(torch.topk(l, 2, dim=-1).values) for l in trace["scores"]]
    m = engine(cfg, sd, "bf16")
    m.post_init_gpt2_config(kv_cache=kv)
    ids, _ = m.inference_speech(None, text, langs=langs, emo_vec=emo, campplus_embedding=style,
                                max_generate_length=int(z["max_gen"]), do_sample=False, num_beams=1, repetition_penalty=10.0)
    firsts = _gated_agreement(ids.cpu().numpy(), ref, margins, BF16_LOGIT_BOUND_SMALL, f"bf16 engine (kv_cache={kv})")
    r16 = z["codes_bf16_kv" if kv else "codes_bf16_nokv"]
    ref_firsts = [int(np.nonzero(r16[r] != ref[r])[0][0]) if (r16[r] != ref[r]).any() else -1 for r in range(ref.shape[0])]
    print(f"first divergence from the fp32 ids per row: engine bf16 {firsts}, the reference's own bf16 mode {ref_firsts}")


# ---- full size: 24 x 1280, B = 64 x 128 text tokens (BASELINE.json configs[2] on one GPU), 64 greedy steps ----------------
@pytest.fixture(scope="module")
def full_size():
    cfg = G.GPTConfig(max_text_tokens=140, max_mel_tokens=200)
    sd = G.synth_weights(cfg, seed=1234)
    sd["mel_head.bias"][cfg.stop_mel_token] -= 1e4                # fixed-length decode
    g = torch.Generator().manual_seed(64)
    B, L, n = 64, 128, 64
    text = torch.randint(2, cfg.number_text_tokens, (B, L), generator=g)
    lens = [L] * B
    for b, nn_ in ((3, 90), (17, 128), (40, 57), (63, 101), (8, 33), (29, 120)):     # ragged rows among full ones
        lens[b] = nn_
        text[b, nn_:] = 1
    style = torch.randn(1, 192, generator=g)
    emo = torch.randn(1, cfg.model_dim, generator=g) * 0.1
    langs = torch.randint(0, cfg.n_langs, (B,), generator=g)
    rows = [3, 17, 40, 63]                                        # rows the CPU oracle decodes
    trace = {}
    with torch.no_grad():
        conds = G.conds_latent_campplus(sd, style, emo)
        sub = text[rows][:, : max(lens[r] for r in rows)]
        oc = G.inference_speech(sd, cfg, conds, sub, langs[rows], G.GenParams(max_generate_length=n), trace=trace)
    margins = [(lambda t: (t[:, 0] - t[:, 1]).numpy())(torch.topk(l, 2, dim=-1).values) for l in trace["scores"]]
    return dict(cfg=cfg, sd=sd, text=text, lens=lens, style=style, emo=emo, langs=langs, rows=rows, oracle_ids=oc.numpy(),
                margins=margins, n=n, conds=conds)


def _decode(m, fs, sel=None, n=None):
    text, langs = fs["text"], fs["langs"]
    if sel is not None:
        text = text[sel][:, : max(fs["lens"][r] for r in sel)]
        langs = langs[sel]
    ids, _ = m.inference_speech(None, text, langs=langs, emo_vec=fs["emo"], campplus_embedding=fs["style"],
                                max_generate_length=n or fs["n"], do_sample=False, num_beams=1, repetition_penalty=10.0)
    return ids.cpu().numpy()


def test_full_size_f32_batch64_ids_vs_oracle_and_row_invariance(full_size):
    """f32 engine at the BASELINE shape: (a) ids of 4 rows of the 64-row device batch == the CPU oracle's ids for those rows
    (64 greedy steps); (b) EVERY row of the 64-row batch equals the same row decoded in a different batch -- 8 rows alone
    (B = 1), the other 56 in 7 groups of 8 -- the reference's tests/padding_test.py:77-99 property at full size."""
    fs = full_size
    m = engine(fs["cfg"], fs["sd"], "fp32")
    full = _decode(m, fs)
    assert full.shape == (64, fs["n"])
    got = full[fs["rows"]]
    if not np.array_equal(got, fs["oracle_ids"]):
        bad = np.argwhere(got != fs["oracle_ids"])[0]
        pytest.fail(f"row {fs['rows'][bad[0]]} step {bad[1]}: engine {got[tuple(bad)]} oracle {fs['oracle_ids'][tuple(bad)]} "
                    f"(fp32 margin there {fs['margins'][bad[1]][bad[0]]:.2e})")
    print("min fp32 top-2 margin (processed scores) over the 4 oracle rows x 64 steps:", min(float(mm.min()) for mm in fs["margins"]))
    singles = [0, 3, 8, 17, 29, 40, 55, 63]
    for r in singles:
        alone = _decode(m, fs, [r])
        assert np.array_equal(alone[0], full[r]), f"row {r} alone != row {r} in the 64-row batch"
    rest = [r for r in range(64) if r not in singles]
    for i in range(0, len(rest), 8):
        grp = rest[i:i + 8]
        part = _decode(m, fs, grp)
        for j, r in enumerate(grp):
            assert np.array_equal(part[j], full[r]), f"row {r} in a group of 8 != row {r} in the 64-row batch"


def test_full_size_bf16_gated_vs_oracle(full_size):
    """The benchmarked mode at the benchmarked size: (a) teacher-forced logits of the bf16 engine on the oracle's ids for 4
    rows within BF16_LOGIT_BOUND_FULL of the CPU oracle's fp32 logits; (b) greedy ids of those rows equal the oracle's
    wherever the fp32 margin exceeds twice that bound (first divergences printed); (c) agreement of all 64 rows with the
    f32 ENGINE's ids reported (prefix lengths), and the bf16 batch is row-invariant like the f32 one."""
    fs = full_size
    cfg, sd, rows, n = fs["cfg"], fs["sd"], fs["rows"], fs["n"]
    m = engine(cfg, sd, "bf16")
    # (a) teacher-forced latents -> logits on the oracle's ids
    tl = torch.tensor([fs["lens"][r] for r in rows])
    sub = fs["text"][rows][:, : int(tl.max())]
    codes = torch.from_numpy(fs["oracle_ids"])
    ml = torch.full((len(rows),), n)
    conds, _ = m.conds_latent(fs["style"], fs["emo"])
    lat = m.forward_latent(conds.repeat(len(rows), 1, 1), sub, tl, codes, ml).cpu()
    with torch.no_grad():
        lat32 = G.forward_latent(sd, cfg, fs["conds"].repeat(len(rows), 1, 1), sub, tl, codes, ml)
    W, b = sd["mel_head.weight"], sd["mel_head.bias"]
    keep = torch.ones(cfg.number_mel_codes, dtype=torch.bool)
    keep[cfg.stop_mel_token] = False                                # the suppressed EOS column carries a -1e4 bias
    d = float((F.linear(_rb(lat), _rb(W), b) - F.linear(lat32, W, b))[..., keep].abs().max())
    print(f"full-size bf16 engine: teacher-forced logits vs CPU fp32 oracle max|d| = {d:.4f} (bound {BF16_LOGIT_BOUND_FULL}); "
          f"latents max|d| = {float((lat - lat32).abs().max()):.4f}")
    assert d <= BF16_LOGIT_BOUND_FULL
    # (b) greedy ids gated by the oracle's margins
    full = _decode(m, fs)
    firsts = _gated_agreement(full[rows], fs["oracle_ids"], fs["margins"], BF16_LOGIT_BOUND_FULL, "full-size bf16 engine")
    print("full-size bf16: first divergence vs the oracle's fp32 ids (4 rows):", firsts)
    # (c) row invariance in bf16: a row decoded alone (LayerNorm-fused 1-row GEMMs, 16-wave attention blocks), in a group of 8 (fused GEMMs on 8
    # waves) and in the 64-row batch (ln_kernel + 64-row slab GEMMs, 4-wave attention blocks) goes through the same sums in the same order --
    # every kernel family keeps one K order and the attention's 16 key streams do not depend on the geometry -- so the ids are EQUAL, not
    # merely close (parity per utterance vs the B = 1 run; the reference's tests/padding_test.py:77-99 property)
    for r in (0, 17, 40):
        alone = _decode(m, fs, [r])
        k = int(np.cumprod(alone[0] == full[r]).sum())
        print(f"bf16 row {r}: alone vs in-batch agreement prefix {k}/{n}")
        assert np.array_equal(alone[0], full[r]), f"bf16 row {r} alone != row {r} in the 64-row batch (first difference at step {k})"
    for grp in ([3, 8, 17, 29, 40, 55, 62, 63], list(range(20, 36))):
        part = _decode(m, fs, grp)
        for j, r in enumerate(grp):
            assert np.array_equal(part[j], full[r]), f"bf16 row {r} in a group of {len(grp)} != row {r} in the 64-row batch"
    m32 = engine(cfg, sd, "fp32")
    ids32 = _decode(m32, fs)
    pref = [int(np.cumprod(full[r] == ids32[r]).sum()) for r in range(64)]
    print(f"bf16 vs f32 ENGINE ids over 64 rows x {n} steps: agreement prefix min/median/max = "
          f"{min(pref)}/{sorted(pref)[32]}/{max(pref)}; rows fully equal: {sum(p == n for p in pref)}")


def test_v2_decode_and_latent_vs_reference_golden(golden_dir):
    """IndexTTS-2 (BASELINE configs[3]) in miniature: `UnifiedVoice` in the reference's default conditioning mode -- 34
    conditioning tokens (32 speaker latents + emo_vec, speed_emb(1), speed_emb(0)), no language embedding -- greedy ids and the
    teacher-forced latent pass of infer_v2.py:636-651 against what the reference's own classes produced (gpt_v2.npz)."""
    from indextts_amd import gpt
    z = np.load(os.path.join(golden_dir, "gpt_v2.npz"))
    c = z["cfg"]
    cfg = G.GPTConfig(layers=int(c[0]), model_dim=int(c[1]), heads=int(c[2]), max_text_tokens=int(c[3]),
                      max_mel_tokens=int(c[4]), number_text_tokens=int(c[5]))
    sd = dict(G.synth_weights(cfg, seed=int(z["seed"])))
    sd["mel_head.bias"][cfg.stop_mel_token] += float(z["eos_bias"])
    sd["speed_emb.weight"] = torch.from_numpy(z["speed_emb"])
    m = gpt.UnifiedVoice(layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, max_text_tokens=cfg.max_text_tokens,
                         max_mel_tokens=cfg.max_mel_tokens, number_text_tokens=cfg.number_text_tokens, precision="fp32", device=DEV)
    assert m.spk_cond_mode == "conformer"                          # the reference default (model_v2.py:312)
    m.load_state_dict(sd)
    m.post_init_gpt2_config(kv_cache=True)
    text, lat = torch.from_numpy(z["text"]), torch.from_numpy(z["spk_latent"])
    B = text.shape[0]
    emo = torch.from_numpy(z["emo_vec"])
    m.conditioning_fn = lambda x, lengths=None: lat.repeat(B, 1, 1).to(DEV)      # stands in for Conformer + Perceiver
    # third positional = `langs`: infer_v2.py:584 passes the emotion features there; ignored outside campplus mode
    codes, spk_out = m.inference_speech(torch.zeros(1, 4, 2), text, torch.zeros(1, 4, 2), emo_vec=emo, cond_lengths=torch.tensor([2]),
                                        max_generate_length=int(z["max_gen"]), do_sample=False, num_beams=1, repetition_penalty=10.0)
    assert np.array_equal(codes.cpu().numpy(), z["codes"])
    assert spk_out.shape == (B, 32, cfg.model_dim)
    out = m(lat.repeat(B, 1, 1), text, torch.from_numpy(z["text_lens"]), torch.from_numpy(z["mel_codes"]), torch.from_numpy(z["mel_lens"]),
            None, emo_vec=emo.repeat(B, 1), use_speed=torch.zeros(B).long(), do_spk_cond=False)
    assert out.shape == z["latent"].shape
    np.testing.assert_allclose(out.cpu().numpy(), z["latent"], rtol=0, atol=5e-5)
