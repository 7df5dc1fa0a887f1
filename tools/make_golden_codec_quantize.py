"""Mint tests/golden/codec_quantize.npz by running the REFERENCE's own `EnhancedCodec.quantize` (indextts/codec/models.py:179-199, the
call of indextts/infer_v2.py:465) on CPU with the oracle's seeded weights (decode half: synth_codec_weights, encoder half:
synth_codec_encoder_weights); prints the oracle-vs-reference agreement.  Import stubs: tools/ref_shim_s2mel.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import codec_oracle as C  # noqa: E402

GOLD = os.path.join(HERE, "..", "tests", "golden")
CFG = C.CodecConfig(codebook_size=96, hidden_size=64, codebook_dim=8, vocos_dim=64, vocos_intermediate_dim=128, vocos_num_layers=3)
SEED = 91
LENGTHS = (21, 34)           # odd and even frame counts -> 11 and 17 codes


def weights():
    sd = C.synth_codec_weights(CFG, SEED)
    sd.update(C.synth_codec_encoder_weights(CFG, SEED + 1))
    return sd


def main():
    import ref_shim_s2mel as R          # the reference is imported only when minting (tests import CFG / LENGTHS / SEED from this module)
    R.install()
    from indextts.codec.models import EnhancedCodec
    sd = weights()
    codec = EnhancedCodec(codebook_size=CFG.codebook_size, hidden_size=CFG.hidden_size, codebook_dim=CFG.codebook_dim, vocos_dim=CFG.vocos_dim,
                          vocos_intermediate_dim=CFG.vocos_intermediate_dim, vocos_num_layers=CFG.vocos_num_layers).eval()
    ref_sd = codec.state_dict()
    for n, shp in C.codec_param_shapes(CFG) + C.codec_encoder_param_shapes(CFG):
        assert n in ref_sd and tuple(ref_sd[n].shape) == shp, (n, shp)
    missing, unexpected = codec.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    g = torch.Generator().manual_seed(SEED + 2)
    out = {}
    for i, T in enumerate(LENGTHS):
        x = torch.randn(2, T, CFG.hidden_size, generator=g)
        with torch.no_grad():
            idx_ref, q_ref = codec.quantize(x)
            idx_o, q_o, margin = C.codec_quantize(sd, CFG, x)
        print(f"quantize T={T}: indices {tuple(idx_ref.shape)} equal {bool(torch.equal(idx_ref, idx_o))}, quantized {tuple(q_ref.shape)} rms "
              f"{q_ref.pow(2).mean().sqrt():.3f} oracle max|d| = {(q_ref - q_o).abs().max():.3e}, smallest search margin {float(margin.min()):.2e}")
        out[f"x{i}"], out[f"idx{i}"], out[f"q{i}"] = x.numpy(), idx_ref.numpy(), q_ref.numpy()
    x1 = torch.randn(1, 9, CFG.hidden_size, generator=g)
    with torch.no_grad():
        idx1, q1 = codec.quantize(x1)                    # batch 1: the reference squeezes the quantizer axis, (1, T')
    print("batch-1 shapes:", tuple(idx1.shape), tuple(q1.shape))
    out["x_b1"], out["idx_b1"], out["q_b1"] = x1.numpy(), idx1.numpy(), q1.numpy()
    np.savez_compressed(os.path.join(GOLD, "codec_quantize.npz"), **out)
    print("wrote codec_quantize.npz")


if __name__ == "__main__":
    main()
