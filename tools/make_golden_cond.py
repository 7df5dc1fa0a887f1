"""Mint tests/golden/cond.npz: the reference's OWN `ConformerEncoder` (indextts/gpt/conformer_encoder.py) and `PerceiverResampler`
(indextts/gpt/perceiver.py), imported from /root/reference, loaded (strict) with oracle/cond_oracle.py's seeded weights and run on
seeded inputs of different lengths -- the composition `UnifiedVoice.get_conditioning` / `get_emo_conditioning` / `get_emovec` /
`merge_emovec` performs (model_v2.py:563-568,588-593,827-838).  The oracle and the engine are tested against these outputs."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import cond_oracle as CO  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
CCFG = CO.ConformerCfg(input_size=36, output_size=64, attention_heads=2, linear_units=128, num_blocks=2)
PCFG = CO.PerceiverCfg(dim=128, dim_context=64, num_latents=8, dim_head=64, heads=2, ff_mult=2.0)
ECFG = CO.ConformerCfg(input_size=36, output_size=64, attention_heads=2, linear_units=96, num_blocks=2)
EPCFG = CO.PerceiverCfg(dim=96, dim_context=64, num_latents=1, dim_head=64, heads=2, ff_mult=2.0)
MODEL_DIM = 128


def weights(seed=11):
    sd = {}
    sd.update(CO.synth_conformer(CCFG, seed, "conditioning_encoder."))
    sd.update(CO.synth_perceiver(PCFG, seed + 1, "perceiver_encoder."))
    sd.update(CO.synth_conformer(ECFG, seed + 2, "emo_conditioning_encoder."))
    sd.update(CO.synth_perceiver(EPCFG, seed + 3, "emo_perceiver_encoder."))
    g = torch.Generator().manual_seed(seed + 4)
    sd["emovec_layer.weight"], sd["emovec_layer.bias"] = torch.randn(MODEL_DIM, EPCFG.dim, generator=g) / EPCFG.dim ** 0.5, 0.05 * torch.randn(MODEL_DIM, generator=g)
    sd["emo_layer.weight"], sd["emo_layer.bias"] = torch.randn(MODEL_DIM, MODEL_DIM, generator=g) / MODEL_DIM ** 0.5, 0.05 * torch.randn(MODEL_DIM, generator=g)
    return sd


def main():
    import importlib.machinery
    import types
    sys.path.insert(0, "/root/reference")
    if "torchaudio" not in sys.modules:                 # indextts/utils/common.py imports it for the WAV helpers only
        stub = types.ModuleType("torchaudio")
        stub.__spec__ = importlib.machinery.ModuleSpec("torchaudio", None)
        sys.modules["torchaudio"] = stub
    from indextts.gpt.conformer_encoder import ConformerEncoder
    from indextts.gpt.perceiver import PerceiverResampler
    sd = weights()

    def sub(pre):
        return {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}

    def conf(cfg, pre):
        m = ConformerEncoder(input_size=cfg.input_size, output_size=cfg.output_size, linear_units=cfg.linear_units,
                             attention_heads=cfg.attention_heads, num_blocks=cfg.num_blocks, input_layer="conv2d2")
        missing, unexpected = m.load_state_dict(sub(pre), strict=False)
        assert not unexpected and all(".pe" in k or "pos_enc" in k for k in missing), (missing, unexpected)
        return m.eval()

    def perc(cfg, pre):
        m = PerceiverResampler(cfg.dim, dim_context=cfg.dim_context, ff_mult=cfg.ff_mult, heads=cfg.heads, num_latents=cfg.num_latents)
        m.load_state_dict(sub(pre), strict=True)
        return m.eval()

    enc, pr = conf(CCFG, "conditioning_encoder."), perc(PCFG, "perceiver_encoder.")
    eenc, epr = conf(ECFG, "emo_conditioning_encoder."), perc(EPCFG, "emo_perceiver_encoder.")
    emovec_layer, emo_layer = nn.Linear(EPCFG.dim, MODEL_DIM), nn.Linear(MODEL_DIM, MODEL_DIM)
    emovec_layer.load_state_dict({"weight": sd["emovec_layer.weight"], "bias": sd["emovec_layer.bias"]})
    emo_layer.load_state_dict({"weight": sd["emo_layer.weight"], "bias": sd["emo_layer.bias"]})
    g = torch.Generator().manual_seed(21)
    B, T = 3, 41
    feats = torch.randn(B, T, CCFG.input_size, generator=g)
    lens = torch.tensor([41, 23, 8])
    emo_feats = torch.randn(2, 29, ECFG.input_size, generator=g)
    emo_lens = torch.tensor([29, 14])
    out = {"feats": feats.numpy(), "lens": lens.numpy(), "emo_feats": emo_feats.numpy(), "emo_lens": emo_lens.numpy()}
    with torch.no_grad():
        h, mask = enc(feats, lens)                                                       # model_v2.py:563-564
        conds = pr(h, nn.ConstantPad1d((PCFG.num_latents, 0), True)(mask.squeeze(1)))    # :567-568
        out.update(enc_out=h.numpy(), enc_mask=mask.numpy(), conds=conds.numpy())

        def emovec(f, l):                                                                # :588-593, 827-831
            hh, mm = eenc(f, l)
            v = epr(hh, nn.ConstantPad1d((1, 0), True)(mm.squeeze(1))).squeeze(1)
            return emo_layer(emovec_layer(v))
        # every prompt ALONE (what the pipeline runs: one prompt per call).  Inside a padded batch the reference's conv module lets
        # GLU(bias of pointwise_conv1) at the PADDED positions into the depthwise conv of a shorter row's last frames
        # (conformer_encoder.py:131-148 zeroes the padding before the biased pointwise conv, not after it), so a shorter row's
        # output depends on how far the batch pads it; the engine (packed rows) computes the alone result for every row.
        for b in range(B):
            n = int(lens[b])
            hb, mb = enc(feats[b:b + 1, :n], lens[b:b + 1])
            cb = pr(hb, nn.ConstantPad1d((PCFG.num_latents, 0), True)(mb.squeeze(1)))
            out[f"enc_out_alone{b}"], out[f"conds_alone{b}"] = hb[0].numpy(), cb[0].numpy()
        ev = emovec(emo_feats, emo_lens)
        for b in range(2):
            n = int(emo_lens[b])
            out[f"emovec_alone{b}"] = emovec(emo_feats[b:b + 1, :n], emo_lens[b:b + 1])[0].numpy()
        base_alone = torch.stack([emovec(feats[b:b + 1, :n], torch.tensor([n]))[0] for b, n in ((0, 29), (1, 23))])
        ev_alone = torch.stack([torch.from_numpy(out[f"emovec_alone{b}"]) for b in range(2)])
        out["merged_alone"] = (base_alone + 0.6 * (ev_alone - base_alone)).numpy()
        base = emovec(feats[:2, :29], torch.tensor([29, 23]))
        out.update(emovec=ev.numpy(), merged=(base + 0.6 * (ev - base)).numpy())
        # oracle check right here
        h2, m2 = CO.conformer_encoder(sd, CCFG, feats, lens, "conditioning_encoder.")
        c2 = CO.conditioning(sd, CCFG, PCFG, feats, lens, "conditioning_encoder.", "perceiver_encoder.")
        e2 = CO.get_emovec(sd, ECFG, EPCFG, emo_feats, emo_lens)
        print("oracle vs reference: encoder", float((h2 - h).abs().max()), "mask equal", bool((m2 == mask).all()), "conds", float((c2 - conds).abs().max()),
              "emovec", float((e2 - ev).abs().max()))
    np.savez_compressed(os.path.join(GOLD, "cond.npz"), **out)
    print("wrote cond.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
