"""GPU parity of the conditioning encoders (Conformer encoder + Perceiver resampler) on the HIP engine,
through the C ABI, against tests/golden/cond.npz = outputs of the REFERENCE's own ConformerEncoder / PerceiverResampler classes
(tools/make_golden_cond.py).  Exact-f32 unit ops: bar 2e-4 absolute on outputs of magnitude ~1."""
import os

import numpy as np
import pytest
import torch

from tools.make_golden_cond import CCFG, ECFG, EPCFG, MODEL_DIM, PCFG, weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-4


def _sub(sd, pre):
    return {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}


def test_attention_unit_vs_torch():
    from indextts_amd.cond import _Ops
    ops = _Ops(DEV)
    g = torch.Generator().manual_seed(1)
    H, dq, dv = 3, 24, 20
    lens = [5, 0, 70, 130]
    nq = [4, 2, 3, 5]
    q = torch.randn(sum(nq), H, dq, generator=g)
    k = torch.randn(sum(lens), H, dq, generator=g)
    v = torch.randn(sum(lens), H, dv, generator=g)
    ks = np.cumsum([0] + lens[:-1])
    kstart = torch.tensor(np.repeat(ks, nq), dtype=torch.int32)
    klen = torch.tensor(np.repeat(lens, nq), dtype=torch.int32)
    out = ops.attention(q.view(-1, H * dq).to(DEV), k.view(-1, H * dq).to(DEV), v.view(-1, H * dv).to(DEV), kstart.to(DEV), klen.to(DEV),
                        H, dq, dv, 0.3).cpu().view(-1, H, dv)
    row = 0
    for b, n in enumerate(nq):
        kk, vv = k[ks[b]: ks[b] + lens[b]], v[ks[b]: ks[b] + lens[b]]
        for i in range(n):
            if lens[b] == 0:
                ref = torch.zeros(H, dv)
            else:
                att = torch.softmax(torch.einsum("hd,jhd->hj", q[row], kk) * 0.3, dim=-1)
                ref = torch.einsum("hj,jhd->hd", att, vv)
            assert float((out[row] - ref).abs().max()) <= 2e-5, (b, i)
            row += 1


def test_conformer_and_perceiver_vs_reference_classes(golden_dir):
    from indextts_amd.cond import ConformerEncoder, PerceiverResampler
    z = np.load(os.path.join(golden_dir, "cond.npz"))
    sd = weights()
    enc = ConformerEncoder(input_size=CCFG.input_size, output_size=CCFG.output_size, linear_units=CCFG.linear_units,
                           attention_heads=CCFG.attention_heads, num_blocks=CCFG.num_blocks, input_layer="conv2d2", device=DEV)
    enc.load_state_dict(_sub(sd, "conditioning_encoder."))
    per = PerceiverResampler(PCFG.dim, dim_context=PCFG.dim_context, ff_mult=PCFG.ff_mult, heads=PCFG.heads, num_latents=PCFG.num_latents, device=DEV)
    per.load_state_dict(_sub(sd, "perceiver_encoder."))
    feats, lens = torch.from_numpy(z["feats"]), torch.from_numpy(z["lens"])
    h, mask = enc(feats, lens)
    assert np.array_equal(mask.cpu().numpy(), z["enc_mask"])
    conds = per(h, torch.nn.functional.pad(mask.squeeze(1), (PCFG.num_latents, 0), value=True))
    assert conds.shape == (3, PCFG.num_latents, PCFG.dim)
    # every row of the ragged batch equals the reference's result for that prompt ALONE (the pipeline's call pattern); the
    # unpadded row also equals the reference's batch result -- shorter rows of a padded reference batch pick up GLU(bias) from
    # their padding in the conv module (see tools/make_golden_cond.py), which the packed engine layout has no rows for
    for b in range(3):
        n = int(mask[b].sum())
        e1 = float((h[b, :n].cpu() - torch.from_numpy(z[f"enc_out_alone{b}"])).abs().max())
        e2 = float((conds[b].cpu() - torch.from_numpy(z[f"conds_alone{b}"])).abs().max())
        print(f"prompt {b} ({n} frames): conformer max|d| {e1:.2e}, perceiver latents max|d| {e2:.2e} vs the reference classes")
        assert e1 <= TOL and e2 <= TOL
    assert float((h[0].cpu() - torch.from_numpy(z["enc_out"][0])).abs().max()) <= TOL
    assert float((conds[0].cpu() - torch.from_numpy(z["conds"][0])).abs().max()) <= TOL


def test_emovec_and_merge_vs_reference(golden_dir):
    from indextts_amd.cond import ConditioningEncoders
    z = np.load(os.path.join(golden_dir, "cond.npz"))
    sd = weights()
    cm = dict(input_size=CCFG.input_size, output_size=CCFG.output_size, linear_units=CCFG.linear_units, attention_heads=CCFG.attention_heads,
              num_blocks=CCFG.num_blocks, input_layer="conv2d2", perceiver_mult=PCFG.ff_mult)
    em = dict(input_size=ECFG.input_size, output_size=ECFG.output_size, linear_units=ECFG.linear_units, attention_heads=ECFG.attention_heads,
              num_blocks=ECFG.num_blocks, input_layer="conv2d2", perceiver_mult=EPCFG.ff_mult, perceiver_dim=EPCFG.dim)
    ce = ConditioningEncoders(MODEL_DIM, cm, em, cond_num=PCFG.num_latents, device=DEV).load_state_dict(sd)
    feats, lens = torch.from_numpy(z["feats"]), torch.from_numpy(z["lens"])
    emo_feats, emo_lens = torch.from_numpy(z["emo_feats"]), torch.from_numpy(z["emo_lens"])
    conds = ce.get_conditioning(feats.transpose(1, 2), lens)
    for b in range(3):
        assert float((conds[b].cpu() - torch.from_numpy(z[f"conds_alone{b}"])).abs().max()) <= TOL
    ev = ce.get_emovec(emo_feats, emo_lens)
    for b in range(2):
        assert float((ev[b].cpu() - torch.from_numpy(z[f"emovec_alone{b}"])).abs().max()) <= TOL
    assert float((ev[0].cpu() - torch.from_numpy(z["emovec"][0])).abs().max()) <= TOL          # the unpadded row of the reference batch
    merged = ce.merge_emovec(feats[:2, :29], emo_feats, torch.tensor([29, 23]), emo_lens, alpha=0.6)
    assert float((merged.cpu() - torch.from_numpy(z["merged_alone"])).abs().max()) <= TOL
    # a prompt alone == the same prompt inside a ragged batch (packed rows: no padding work, no leakage)
    alone = ce.get_conditioning(feats[1:2, :23].transpose(1, 2), torch.tensor([23]))
    assert float((alone - conds[1:2]).abs().max()) <= 1e-5


def test_unified_voice_uses_engine_encoders(golden_dir):
    """`UnifiedVoice(**cfg.gpt)` with the reference's `condition_module` / `emo_condition_module` sections and a checkpoint that
    carries the encoder weights: `get_conditioning`, `get_emovec`, `merge_emovec` run on the engine (no `conditioning_fn`), and
    `inference_speech(speech_condition, ...)` decodes from the engine-computed 32 (+2) conditioning tokens exactly like a call
    that is handed the same latents."""
    from indextts_amd import gpt
    from oracle import gpt_oracle as G
    z = np.load(os.path.join(golden_dir, "cond.npz"))
    cfg = G.GPTConfig(layers=2, model_dim=MODEL_DIM, heads=2, max_text_tokens=40, max_mel_tokens=40, number_text_tokens=100)
    sd = dict(G.synth_weights(cfg, seed=5))
    sd["speed_emb.weight"] = torch.randn(2, MODEL_DIM, generator=torch.Generator().manual_seed(6)) * 0.2
    sd.update(weights())
    cm = dict(input_size=CCFG.input_size, output_size=CCFG.output_size, linear_units=CCFG.linear_units, attention_heads=CCFG.attention_heads,
              num_blocks=CCFG.num_blocks, input_layer="conv2d2", perceiver_mult=PCFG.ff_mult)
    em = dict(input_size=ECFG.input_size, output_size=ECFG.output_size, linear_units=ECFG.linear_units, attention_heads=ECFG.attention_heads,
              num_blocks=ECFG.num_blocks, input_layer="conv2d2", perceiver_mult=EPCFG.ff_mult, perceiver_dim=EPCFG.dim)
    m = gpt.UnifiedVoice(layers=2, model_dim=MODEL_DIM, heads=2, max_text_tokens=40, max_mel_tokens=40, number_text_tokens=100,
                         condition_type="conformer_perceiver", condition_num_latent=PCFG.num_latents, condition_module=cm,
                         emo_condition_module=em, precision="fp32", device=DEV)
    ignored = m.load_state_dict(sd)
    assert m.cond_encoders is not None and not [k for k in ignored if "conditioning_encoder" in k or "perceiver" in k or "emo" in k]
    feats = torch.from_numpy(z["feats"])[:1]                                        # (1, 41, 36) prompt features
    lat = m.get_conditioning(feats.transpose(1, 2), torch.tensor([41]))
    assert float((lat[0].cpu() - torch.from_numpy(z["conds_alone0"])).abs().max()) <= TOL
    emo = m.merge_emovec(feats[:, :29], torch.from_numpy(z["emo_feats"])[:1], torch.tensor([29]), torch.tensor([29]), alpha=0.6)
    assert float((emo[0].cpu() - torch.from_numpy(z["merged_alone"][0])).abs().max()) <= TOL
    text = torch.randint(2, 100, (1, 12), generator=torch.Generator().manual_seed(7))
    with pytest.raises(ValueError, match="ConformerEncoder"):                     # channel-first features where (B, T, 1024) is expected
        m.inference_speech(feats.transpose(1, 2), text, emo_vec=emo, cond_lengths=torch.tensor([41]), max_generate_length=4)
    a, _ = m.inference_speech(feats, text, emo_vec=emo, cond_lengths=torch.tensor([41]), max_generate_length=10,
                              do_sample=False, num_beams=1, repetition_penalty=10.0)
    b, _ = m.inference_speech(None, text, emo_vec=emo, conds_latent=m.conds_latent_v2(lat, emo), max_generate_length=10, do_sample=False,
                              num_beams=1, repetition_penalty=10.0)
    assert torch.equal(a, b) and a.shape[1] >= 1
