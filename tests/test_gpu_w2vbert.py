"""GPU parity of the w2v-bert-2.0 feature encoder (`semantic_model` -> `get_emb`) on the HIP engine, through the
C ABI: unit ops against torch, the small model against tests/golden/w2vbert.npz (transformers' own Wav2Vec2BertModel on the oracle's
seeded weights), and the full-width pipeline configuration (hidden 1024, 16 heads, FFN 4096, 17 layers to the tapped state) against
the oracle.  Exact-f32 unit ops; LayerNorm after every layer keeps the activations O(1): bar 2e-4 absolute (small), 1e-3 (17 layers)."""
import math
import os
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import w2vbert_oracle as WO
from tools.make_golden_w2vbert import CFG, LAYER

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_relkey_attention_and_causal_dwconv_vs_torch():
    from indextts_amd.codec import _tables
    from indextts_amd.w2vbert import _WOps
    ops = _WOps(DEV)
    g = torch.Generator().manual_seed(5)
    lens, H, dh, left, right = [70, 0, 133], 3, 32, 9, 4
    (tok_seq, tok_t, start, Tt), n = _tables(lens, DEV)
    q, k, v = (torch.randn(n, H * dh, generator=g) for _ in range(3))
    dist = torch.randn(left + right + 1, dh, generator=g)
    kstart, klen = start[tok_seq.long()].contiguous(), Tt[tok_seq.long()].contiguous()
    out = ops.attention_relkey(q.to(DEV), k.to(DEV), v.to(DEV), kstart, klen, tok_t, dist.to(DEV), left, right, H, dh, dh, 1 / math.sqrt(dh)).cpu()
    s0 = 0
    for T in lens:
        if T == 0:
            continue
        qq, kk, vv = (t[s0:s0 + T].view(T, H, dh).transpose(0, 1) for t in (q, k, v))
        d = (torch.arange(T)[None, :] - torch.arange(T)[:, None]).clamp(-left, right) + left
        sc = (qq @ kk.transpose(1, 2) + torch.einsum("hld,lrd->hlr", qq, dist[d])) / math.sqrt(dh)
        ref = (torch.softmax(sc, -1) @ vv).transpose(0, 1).reshape(T, H * dh)
        assert float((out[s0:s0 + T] - ref).abs().max()) <= 2e-5
        s0 += T
    from indextts_amd.gpt import layernorm
    for D in (160, 36):                                           # widths off the 64-lane grid (the 160-wide stacked fbank rows)
        xx, gg, bb = torch.randn(77, D, generator=g) * 3 + 1, torch.randn(D, generator=g), torch.randn(D, generator=g)
        y = layernorm(xx.to(DEV), gg.to(DEV), bb.to(DEV), eps=1e-5).cpu()
        assert float((y - F.layer_norm(xx, (D,), gg, bb, 1e-5)).abs().max()) <= 1e-5
    C, kk = 48, 7
    x, w = torch.randn(n, C, generator=g), torch.randn(C, kk, generator=g)
    y = ops.dwconv_causal(x.to(DEV), w.to(DEV), tok_seq, tok_t, Tt, kk).cpu()
    s0 = 0
    for T in lens:
        if T:
            ref = F.conv1d(F.pad(x[s0:s0 + T].t().unsqueeze(0), (kk - 1, 0)), w.unsqueeze(1), groups=C)[0].t()
            assert float((y[s0:s0 + T] - ref).abs().max()) <= 1e-5
        s0 += T


def test_small_model_vs_transformers_class(golden_dir):
    from indextts_amd.w2vbert import Wav2Vec2BertModel
    z = np.load(os.path.join(golden_dir, "w2vbert.npz"))
    m = Wav2Vec2BertModel(**CFG.__dict__, device=DEV).load_state_dict(WO.synth_weights(CFG))
    feats, mask = torch.from_numpy(z["feats"]), torch.from_numpy(z["mask"])
    valid = mask.bool()
    emb = m.get_emb(feats, mask, torch.from_numpy(z["mean"]), torch.from_numpy(z["std"]), layer=LAYER).cpu()
    e1 = float((emb - torch.from_numpy(z["emb"]))[valid].abs().max())
    out = m(feats.to(DEV), mask.to(DEV), output_hidden_states=True)
    e2 = float((out.hidden_states[1].cpu() - torch.from_numpy(z["h1"]))[valid].abs().max())
    e3 = float((out.last_hidden_state.cpu() - torch.from_numpy(z["last"]))[valid].abs().max())
    print(f"w2v-bert small: max|d| vs transformers  emb {e1:.2e}  h1 {e2:.2e}  last {e3:.2e}")
    assert max(e1, e2, e3) <= 2e-4
    assert float(out.last_hidden_state.cpu()[~valid].abs().max()) == 0.0
    with pytest.raises(ValueError):
        m(torch.zeros(1, 10, CFG.feature_projection_input_dim + 1))


def test_pipeline_width_17_layers_vs_oracle():
    from indextts_amd.w2vbert import Wav2Vec2BertModel
    cfg = WO.W2VBertCfg(num_hidden_layers=17)                      # the layers below hidden_states[17]; widths of facebook/w2v-bert-2.0
    sd = WO.synth_weights(cfg, seed=41)
    m = Wav2Vec2BertModel(**cfg.__dict__, device=DEV).load_state_dict(sd)
    g = torch.Generator().manual_seed(43)
    T = 160
    feats = torch.randn(2, T, 160, generator=g)
    mask = torch.ones(2, T, dtype=torch.long)
    mask[1, 101:] = 0
    mean, std = torch.zeros(1024), torch.ones(1024)
    ref = WO.get_emb(sd, cfg, feats, mask, mean, std, layer=17)
    emb = m.get_emb(feats, mask, mean, std, layer=17).cpu()
    err = float((emb - ref)[mask.bool()].abs().max())
    print(f"w2v-bert 1024 x 17 layers: max|d| vs the oracle {err:.2e} (rms {float(ref.pow(2).mean().sqrt()):.2f})")
    assert err <= 1e-3
    feats15 = torch.randn(1, 749, 160, generator=g).to(DEV)       # a 15 s prompt (the pipeline's cut, infer_v2_5.py:630)
    m15 = torch.ones(1, 749, dtype=torch.long, device=DEV)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.get_emb(feats15, m15, mean, std, layer=17)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"w2v-bert get_emb, 15 s prompt (749 frames, 17 layers): {dt * 1e3:.1f} ms")
