"""CPU tests of the host side: the C ABI loads and exports every declared symbol (no compute calls), the pure-CPU
weight packers, and the torch input-assembly mirror of `prepare_gpt_inputs` against the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import gpt_oracle as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_header_symbol_is_exported_and_bound():
    from indextts_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "indextts_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(itts_[A-Za-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/indextts_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) <= declared
    assert L.itts_abi_version() == _lib.ABI_VERSION


def test_product_path_fails_loudly_without_device():
    from indextts_amd import _lib, bigvgan
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.HipEngineError):
        bigvgan.anti_alias_activation(torch.zeros(1, 1, 4), torch.zeros(12), torch.zeros(12), torch.zeros(1), torch.zeros(1))


def test_conv_packing_layout():
    from indextts_amd import bigvgan
    g = torch.Generator().manual_seed(0)
    Cout, Cin, k = 40, 16, 3
    w = torch.randn(Cout, Cin, k, generator=g)
    p = bigvgan.pack_conv1d_weight(w).view(2, k, Cin // 8, 64, 4)
    for cs, j, c8, lane, s in [(0, 0, 0, 0, 0), (1, 2, 1, 37, 3), (0, 1, 1, 63, 2), (1, 0, 0, 7, 1)]:
        co, ci = cs * 32 + (lane & 31), c8 * 8 + 2 * s + (lane >> 5)
        want = float(w[co, ci, j]) if co < Cout else 0.0
        assert float(p[cs, j, c8, lane, s]) == want
    assert float(p[1, 0, 0, 8, 0]) == 0.0 and float(p[1, 0, 0, 7, 0]) == float(w[39, 0, 0])    # co 40 is padding
    # transposed conv phases: tap jj of phase r is kernel index r + jj*u
    u, kt = 2, 4
    wt = torch.randn(Cin, 24, kt, generator=g)
    pt = bigvgan.pack_convT_weight(wt, u).view(u, 1, 2, Cin // 8, 64, 4)
    for r, jj, c8, lane, s in [(0, 0, 0, 5, 0), (1, 1, 1, 40, 2)]:
        co, ci = lane & 31, c8 * 8 + 2 * s + (lane >> 5)
        want = float(wt[ci, co, r + jj * u]) if co < 24 else 0.0
        assert float(pt[r, 0, jj, c8, lane, s]) == want


@pytest.mark.parametrize("prec", [0, 1])
def test_gemm_packing_layout(prec):
    from indextts_amd import gpt
    g = torch.Generator().manual_seed(1)
    K, N = 64, 40
    w = torch.randn(K, N, generator=g)
    raw = gpt.pack_gemm_weight(w, prec)
    KB, per = (32, 8) if prec == 1 else (16, 4)
    if prec == 1:
        p = raw.view(torch.bfloat16).view(3, K // KB, 64, per).float()
        ref = w.bfloat16().float()
    else:
        p = raw.view(torch.float32).view(3, K // KB, 64, per)
        ref = w
    for nt, kb, lane, j in [(0, 0, 0, 0), (2, 1, 63, per - 1), (1, 0, 17, 2), (2, 0, 9, 1)]:
        n, k = nt * 16 + (lane & 15), kb * KB + (lane >> 4) * per + j
        want = float(ref[k, n]) if n < N else 0.0
        assert float(p[nt, kb, lane, j]) == want
    # nn.Linear layout [N][K]
    raw_t = gpt.pack_gemm_weight(w.t().contiguous(), prec, transposed=True)
    assert torch.equal(raw, raw_t)


def _host_model(cfg, sd):
    from indextts_amd import gpt
    m = gpt.UnifiedVoice(spk_cond_mode="campplus", layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, max_text_tokens=cfg.max_text_tokens,
                         max_mel_tokens=cfg.max_mel_tokens, number_text_tokens=cfg.number_text_tokens, device="cpu",
                         precision="fp32")
    for n in m._HOST_TENSORS:
        m._emb[n] = sd[n]
    return m


def test_prepare_gpt_inputs_matches_oracle():
    cfg = G.GPTConfig(layers=1, model_dim=64, heads=1, max_text_tokens=30, max_mel_tokens=40, number_text_tokens=100)
    sd = G.synth_weights(cfg, seed=3)
    m = _host_model(cfg, sd)
    g = torch.Generator().manual_seed(4)
    text = torch.randint(2, 100, (5, 12), generator=g)
    text[1, 7:] = 1          # right padded with stop tokens
    text[2, :3] = 0          # stray start tokens on the left are stripped too
    text[3, 5] = 1           # a stop token in the middle is stripped
    text[4, :] = 1           # empty row: only [start, stop] remain
    style = torch.randn(1, 192, generator=g)
    emo = torch.randn(1, 64, generator=g)
    langs = torch.randint(0, cfg.n_langs, (5,), generator=g)
    conds = G.conds_latent_campplus(sd, style, emo)
    conds_m, _ = m.conds_latent(style, emo)
    assert torch.allclose(conds, conds_m, atol=1e-6)
    fo, eo, mo = G.prepare_gpt_inputs(sd, cfg, conds, text, langs)
    fm, em, mm = m.prepare_gpt_inputs(conds_m, text, langs)
    assert torch.equal(fo, fm) and torch.equal(mo, mm)
    assert torch.allclose(eo, em, atol=1e-6)
    # per-row conditioning and a single shared language id
    condsB = conds.repeat(5, 1, 1) + torch.randn(5, 1, 64, generator=g)
    fo, eo, mo = G.prepare_gpt_inputs(sd, cfg, condsB, text, langs[:1])
    fm, em, mm = m.prepare_gpt_inputs(condsB, text, langs[:1])
    assert torch.equal(mo, mm) and torch.allclose(eo, em, atol=1e-6)


# ---- v1 pipeline helpers vs the reference's own functions (fixture: tools/make_golden_host.py) ----------------------
def _v1_shell(version=None):
    from indextts_amd.infer import IndexTTS
    t = IndexTTS.__new__(IndexTTS)                       # helpers need no engine / GPU
    t.stop_mel_token = 8193
    t.cfg = {"gpt": {"stop_text_token": 1, "start_text_token": 0}}
    t.model_version = version
    return t


def _host_golden(golden_dir):
    import json
    import os
    with open(os.path.join(golden_dir, "host_v1.json")) as f:
        return json.load(f)


def test_v1_remove_long_silence_matches_reference(golden_dir):
    import torch
    t = _v1_shell()
    for case in _host_golden(golden_dir)["silence"]:
        codes, lens = t.remove_long_silence(torch.tensor(case["codes"], dtype=torch.long), silent_token=52, max_consecutive=30)
        assert codes.tolist() == case["out"]
        assert lens.tolist() == case["lens"]


def test_v1_bucket_segments_matches_reference(golden_dir):
    t = _v1_shell()
    for case in _host_golden(golden_dir)["buckets"]:
        segs = [["t"] * n for n in case["lens"]]
        res = t.bucket_segments(segs, bucket_max_size=case["bucket_max_size"])
        assert [[it["idx"] for it in b] for b in res] == case["out"], case


def test_v1_pad_tokens_cat_matches_reference(golden_dir):
    import torch
    for case in _host_golden(golden_dir)["pad"]:
        t = _v1_shell(case["version"])
        res = t.pad_tokens_cat([torch.tensor(x) for x in case["tokens"]])
        assert res.tolist() == case["out"]


def test_golden_tools_import_without_the_reference():
    """tests import constants (shapes, seeds) from tools/make_golden_*.py; /root/reference does not exist on the GPU box, so importing those
    modules must not touch the reference package -- only their main() may."""
    import glob
    import re
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    used = set()
    for f in glob.glob(os.path.join(root, "tests", "*.py")):
        used.update(re.findall(r"from tools\.(make_golden_\w+) import", open(f).read()))
    assert used
    code = "import sys; sys.modules['indextts'] = None; sys.path[:] = [p for p in sys.path if 'reference' not in p]\n" + \
           "\n".join(f"import tools.{m}" for m in sorted(used)) + "\nassert not any('reference' in p for p in sys.path), sys.path"
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_x3_plane_layout_is_the_fragment_order_of_the_tile_kernel():
    """CPU restatement of the layout `ada_rmsnorm_planes_kernel` writes and `gemm_x3_kernel<..., APL>` reads (s2mel_kernels.hip / gpt_kernels.hip): inside
    every 32-column group the producer puts column c at offset  8 * ((c & 15) >> 2) + 4 * (c >> 4) + (c & 3)  -- a bijection of 0..31 -- so that the
    16-byte piece kg holds, in order, columns 4 kg .. 4 kg + 3 and 16 + 4 kg .. + 3: the eight k-values the f32 tile kernel's lane (row, kg) reads as
    its two f32 pieces and x3_split8 packs into one MFMA operand.  And the LDS image: DMA lane l of a 16-row chunk carries (row l >> 2, source piece
    (l & 3) ^ ((l >> 4) & 3)); the reader of (row16, kg) looks in slot kg ^ ((row16 >> 2) & 3) and must find source piece kg, and the sixteen lanes
    of one k-group must touch sixteen different 16-byte bank quads of the 1 KiB chunk (conflict-free ds_read_b128)."""
    off = [8 * ((c & 15) >> 2) + 4 * (c >> 4) + (c & 3) for c in range(32)]
    assert sorted(off) == list(range(32))
    inv = {o: c for c, o in enumerate(off)}
    for kg in range(4):
        assert [inv[8 * kg + j] for j in range(8)] == [4 * kg + j for j in range(4)] + [16 + 4 * kg + j for j in range(4)]
    lds = {}                                                   # slot index inside the chunk -> (row, source piece)
    for lane in range(64):
        lds[lane] = (lane >> 2, (lane & 3) ^ ((lane >> 4) & 3))
    for kg in range(4):
        quads = set()
        for row16 in range(16):
            slot = row16 * 4 + (kg ^ ((row16 >> 2) & 3))
            assert lds[slot] == (row16, kg)
            quads.add(slot % 16)
        assert len(quads) == 16
