"""GPU parity tests of the codes -> content-feature step on the HIP engine: `EnhancedCodec.decode` and the s2mel
`InterpolateRegulator`, against `tests/golden/codec_e64.npz` -- outputs of the REFERENCE's own classes on the oracle's seeded
weights (tools/make_golden_codec.py) -- and against torch for the unit ops.  Bar: 1e-4 absolute (f32 arithmetic throughout)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import codec_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "codec_e64.npz"))
    a, b = [int(v) for v in z["codec_cfg"]], [int(v) for v in z["reg_cfg"]]
    cc = O.CodecConfig(codebook_size=a[0], hidden_size=a[1], codebook_dim=a[2], vocos_dim=a[3], vocos_intermediate_dim=a[4], vocos_num_layers=a[5])
    rc = O.RegulatorConfig(channels=b[0], in_channels=b[1], n_layers=b[2], groups=b[3], codebook_size=b[4])
    seed = int(z["seed"])
    return z, cc, rc, O.synth_codec_weights(cc, seed), O.synth_regulator_weights(rc, seed + 1)


def engines(cc, rc, csd, rsd):
    from indextts_amd import codec
    c = codec.EnhancedCodec(codebook_size=cc.codebook_size, hidden_size=cc.hidden_size, codebook_dim=cc.codebook_dim, vocos_dim=cc.vocos_dim,
                            vocos_intermediate_dim=cc.vocos_intermediate_dim, vocos_num_layers=cc.vocos_num_layers, device=DEV)
    c.load_state_dict(csd)
    r = codec.InterpolateRegulator(channels=rc.channels, sampling_ratios=(1,) * rc.n_layers, is_discrete=False, in_channels=rc.in_channels,
                                   codebook_size=rc.codebook_size, device=DEV)
    r.load_state_dict(rsd)
    return c, r


def test_unit_ops_vs_torch():
    from indextts_amd import codec
    ops = codec._TokOps(DEV)
    g = torch.Generator().manual_seed(1)
    lens = [9, 1, 30]
    (tok_seq, tok_t, start, T), n = codec._tables(lens, DEV)
    Cc = 64
    x = torch.randn(n, Cc, generator=g)
    # depthwise conv k=7, zero padded at each sequence's own ends
    w, b = torch.randn(Cc, 7, generator=g), torch.randn(Cc, generator=g)
    y = ops.dwconv(x.to(DEV), w.to(DEV), b.to(DEV), tok_seq, tok_t, T, 7).cpu()
    o = 0
    for L in lens:
        ref = F.conv1d(x[o:o + L].t()[None], w[:, None, :], b, padding=3, groups=Cc)[0].t()
        assert (y[o:o + L] - ref).abs().max() < 1e-5
        o += L
    # nearest interpolation + conv k=3 (regulator layer 0) and x2 upsampling + conv (codec `up`)
    wc, bc = torch.randn(32, Cc, 3, generator=g) / 14, torch.randn(32, generator=g)
    wp = codec.pack_gemm_weight(codec._conv_matrix(wc), 0).to(DEV)
    for dst in ([15, 2, 47], [18, 2, 60], lens):
        (dseq, dt, dstart, dT), nd = codec._tables(dst, DEV)
        y = ops.conv(x.to(DEV), (dseq, dt), nd, start, T, dT, 3, wp, bc.to(DEV), 32).cpu()
        o, od = 0, 0
        for L, Ld in zip(lens, dst):
            xi = F.interpolate(x[o:o + L].t()[None], size=Ld, mode="nearest")
            ref = F.conv1d(xi, wc, bc, padding=1)[0].t()
            assert (y[od:od + Ld] - ref).abs().max() < 2e-5, (dst, L, Ld)
            o, od = o + L, od + Ld
    # GroupNorm(1) + Mish per sequence, exact GELU, scale-residual
    gm, bt = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    h = ops.groupnorm_mish_(x.clone().to(DEV), gm.to(DEV), bt.to(DEV), start, T).cpu()
    o = 0
    for L in lens:
        ref = F.mish(F.group_norm(x[o:o + L].t()[None], 1, gm, bt, 1e-5))[0].t()
        assert (h[o:o + L] - ref).abs().max() < 2e-5
        o += L
    assert (ops.gelu_(x.clone().to(DEV)).cpu() - F.gelu(x)).abs().max() < 1e-6
    y2 = torch.randn(n, Cc, generator=g)
    assert (ops.scale_residual_(x.clone().to(DEV), y2.to(DEV), gm.to(DEV)).cpu() - (x + gm * y2)).abs().max() < 1e-6


def test_codec_decode_and_regulator_vs_reference(golden_dir):
    z, cc, rc, csd, rsd = load(golden_dir)
    c, r = engines(cc, rc, csd, rsd)
    codes = torch.from_numpy(z["codes"])
    s = c.decode(codes.to(DEV)).cpu()
    assert s.shape == z["s_infer"].shape
    e1 = float((s - torch.from_numpy(z["s_infer"])).abs().max())
    cond, olens, *_ = r(torch.from_numpy(z["s_infer"]).to(DEV), ylens=torch.from_numpy(z["ylens"]), n_quantizers=3, f0=None)
    e2 = float((cond.cpu() - torch.from_numpy(z["cond"])).abs().max())
    print(f"codec.decode max|d| vs reference {e1:.2e}; length_regulator (reference batch semantics) max|d| {e2:.2e}")
    assert e1 <= TOL and e2 <= TOL
    assert olens.tolist() == z["ylens"].tolist()
    assert float(cond[1, int(z["ylens"][1]):].abs().max()) == 0.0


def test_ragged_batch_equals_per_utterance(golden_dir):
    """New capability: utterances of different code lengths in one call; each row equals its own batch-1 result (what the
    reference computes per utterance) -- decode with code_lens, regulator with frame_lens = ylens."""
    z, cc, rc, csd, rsd = load(golden_dir)
    c, r = engines(cc, rc, csd, rsd)
    g = torch.Generator().manual_seed(9)
    lens = [21, 8, 15]
    codes = torch.randint(0, cc.codebook_size, (3, 21), generator=g)
    with torch.no_grad():
        refs = [O.codec_decode(csd, cc, codes[b:b + 1, :n]) for b, n in enumerate(lens)]
    s = c.decode(codes.to(DEV), code_lens=lens).cpu()
    for b, n in enumerate(lens):
        assert float((s[b:b + 1, : 2 * n] - refs[b]).abs().max()) <= TOL
        assert float(s[b, 2 * n:].abs().max()) == 0.0 if 2 * n < s.shape[1] else True
    ylens = [int(2 * n * 1.72) for n in lens]
    cond, _, *_ = r(s.to(DEV), ylens=torch.tensor(ylens), xlens=[2 * n for n in lens], frame_lens=ylens)
    for b, n in enumerate(lens):
        with torch.no_grad():
            ref, _ = O.length_regulator(rsd, rc, refs[b], torch.tensor([ylens[b]]))
        assert float((cond[b:b + 1, : ylens[b]].cpu() - ref).abs().max()) <= TOL


def test_quantize_vs_reference_class(golden_dir):
    """`EnhancedCodec.quantize` on the engine (stride-2 down conv, Vocos encoder, FVQ search, out_project) vs tests/golden/codec_quantize.npz =
    the reference's own class on the oracle's seeded weights (tools/make_golden_codec_quantize.py).  Indices must be identical wherever the
    oracle's search margin (best minus second-best negative distance) exceeds 1e-4 -- every position of the fixture (smallest margin 6e-3);
    quantized features within 1e-4 where the index agrees."""
    from indextts_amd import codec
    from tools.make_golden_codec_quantize import CFG, LENGTHS, SEED
    z = np.load(os.path.join(golden_dir, "codec_quantize.npz"))
    sd = O.synth_codec_weights(CFG, SEED)
    sd.update(O.synth_codec_encoder_weights(CFG, SEED + 1))
    c = codec.EnhancedCodec(codebook_size=CFG.codebook_size, hidden_size=CFG.hidden_size, codebook_dim=CFG.codebook_dim, vocos_dim=CFG.vocos_dim,
                            vocos_intermediate_dim=CFG.vocos_intermediate_dim, vocos_num_layers=CFG.vocos_num_layers, device=DEV)
    assert c.load_state_dict(sd) == []
    for i, T in enumerate(LENGTHS):
        x = torch.from_numpy(z[f"x{i}"])
        idx, q = c.quantize(x.to(DEV))
        _, _, margin = O.codec_quantize(sd, CFG, x)
        idx, q = idx.cpu(), q.cpu()
        same = idx == torch.from_numpy(z[f"idx{i}"])
        print(f"quantize T={T}: {int(same.sum())}/{same.numel()} indices equal the reference's, smallest margin {float(margin.min()):.2e}, "
              f"quantized max|d| {float((q - torch.from_numpy(z[f'q{i}']))[same].abs().max()):.2e}")
        assert idx.shape == z[f"idx{i}"].shape and idx.dtype == torch.int64 and bool(same[margin > 1e-4].all())
        assert float((q - torch.from_numpy(z[f"q{i}"]))[same].abs().max()) <= TOL
    # batch 1 keeps the (1, T') index shape of the reference's squeeze; ragged rows equal the rows run alone
    idx1, q1 = c.quantize(torch.from_numpy(z["x_b1"]).to(DEV))
    assert idx1.shape == z["idx_b1"].shape and torch.equal(idx1.cpu(), torch.from_numpy(z["idx_b1"]))
    assert float((q1.cpu() - torch.from_numpy(z["q_b1"])).abs().max()) <= TOL
    x = torch.from_numpy(z["x1"])
    idx_r, q_r = c.quantize(x.to(DEV), lens=[34, 21])
    ia, qa = c.quantize(x[1:2, :21].to(DEV))
    assert torch.equal(idx_r[1, :11], ia[0]) and float((q_r[1, :11] - qa[0]).abs().max()) <= 1e-6 and float(q_r[1, 11:].abs().max()) == 0.0
    # a decode-only checkpoint refuses quantize loudly
    d = codec.EnhancedCodec(codebook_size=CFG.codebook_size, hidden_size=CFG.hidden_size, codebook_dim=CFG.codebook_dim, vocos_dim=CFG.vocos_dim,
                            vocos_intermediate_dim=CFG.vocos_intermediate_dim, vocos_num_layers=CFG.vocos_num_layers, device=DEV)
    d.load_state_dict(O.synth_codec_weights(CFG, SEED))
    with pytest.raises(RuntimeError):
        d.quantize(x.to(DEV))


def test_v2_prompt_condition_chain(golden_dir):
    """IndexTTS-2's prompt content condition (indextts/infer_v2.py:465-479): quantize(spk_cond_emb) -> length_regulator, engine vs the oracle chain."""
    from indextts_amd import codec
    from tools.make_golden_codec_quantize import CFG, SEED
    z = np.load(os.path.join(golden_dir, "codec_quantize.npz"))
    sd = O.synth_codec_weights(CFG, SEED)
    sd.update(O.synth_codec_encoder_weights(CFG, SEED + 1))
    rc = O.RegulatorConfig(channels=64, in_channels=CFG.hidden_size, n_layers=4, groups=1, codebook_size=64)
    rsd = O.synth_regulator_weights(rc, SEED + 5)
    c = codec.EnhancedCodec(codebook_size=CFG.codebook_size, hidden_size=CFG.hidden_size, codebook_dim=CFG.codebook_dim, vocos_dim=CFG.vocos_dim,
                            vocos_intermediate_dim=CFG.vocos_intermediate_dim, vocos_num_layers=CFG.vocos_num_layers, device=DEV)
    c.load_state_dict(sd)
    r = codec.InterpolateRegulator(channels=rc.channels, sampling_ratios=(1,) * rc.n_layers, is_discrete=False, in_channels=rc.in_channels,
                                   codebook_size=rc.codebook_size, device=DEV)
    r.load_state_dict(rsd)
    x = torch.from_numpy(z["x_b1"])
    ylens = torch.tensor([23])
    _, S_ref = c.quantize(x.to(DEV))
    cond = r(S_ref, ylens=ylens, n_quantizers=3, f0=None)[0].cpu()
    _, q_o, _ = O.codec_quantize(sd, CFG, x)
    ref, _ = O.length_regulator(rsd, rc, q_o, ylens)
    assert cond.shape == ref.shape == (1, 23, 64) and float((cond - ref).abs().max()) <= TOL
