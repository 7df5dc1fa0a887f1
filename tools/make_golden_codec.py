"""Mint tests/golden/codec.npz by running the REFERENCE's own `EnhancedCodec.decode` (indextts/codec/models.py) and
`InterpolateRegulator` (indextts/s2mel/modules/length_regulator.py) on CPU with the oracle's seeded weights; prints the
oracle-vs-reference agreement.  Import stubs: tools/ref_shim_s2mel.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
import ref_shim_s2mel as R  # noqa: E402

R.install()
from indextts.codec.models import EnhancedCodec  # noqa: E402
from indextts.s2mel.modules.length_regulator import InterpolateRegulator  # noqa: E402

from oracle import codec_oracle as C  # noqa: E402

GOLD = os.path.join(HERE, "..", "tests", "golden")


def main(tag="codec", cc=None, rc=None, seed=71, n_codes=13):
    cc = cc or C.CodecConfig(codebook_size=64, hidden_size=32, codebook_dim=8, vocos_dim=24, vocos_intermediate_dim=48, vocos_num_layers=3)
    rc = rc or C.RegulatorConfig(channels=16, in_channels=32, n_layers=4, groups=1, codebook_size=64)
    csd, rsd = C.synth_codec_weights(cc, seed), C.synth_regulator_weights(rc, seed + 1)
    codec = EnhancedCodec(codebook_size=cc.codebook_size, hidden_size=cc.hidden_size, codebook_dim=cc.codebook_dim, vocos_dim=cc.vocos_dim,
                          vocos_intermediate_dim=cc.vocos_intermediate_dim, vocos_num_layers=cc.vocos_num_layers).eval()
    ref_sd = codec.state_dict()
    for n, shp in C.codec_param_shapes(cc):
        assert n in ref_sd and tuple(ref_sd[n].shape) == shp, (n, shp)
    missing, unexpected = codec.load_state_dict(csd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith(("encoder", "down", "quantizer.quantizers.0.in_project")) for k in missing), missing
    reg = InterpolateRegulator(channels=rc.channels, sampling_ratios=(1,) * rc.n_layers, is_discrete=False, in_channels=rc.in_channels,
                               vector_quantize=False, codebook_size=rc.codebook_size, out_channels=rc.channels, groups=rc.groups,
                               n_codebooks=1, quantizer_dropout=0.0, f0_condition=False, n_f0_bins=512).eval()
    assert [k for k in reg.state_dict()] == [n for n, _ in C.regulator_param_shapes(rc)]
    reg.load_state_dict(rsd, strict=True)
    g = torch.Generator().manual_seed(seed + 2)
    codes = torch.randint(0, cc.codebook_size, (2, n_codes), generator=g)
    with torch.no_grad():
        s_ref = codec.decode(codes)                                            # (2, 26, 32)
        s_o = C.codec_decode(csd, cc, codes)
        ylens = torch.tensor([int(s_ref.shape[1] * 1.72), int(s_ref.shape[1] * 1.72 * 0.6)])
        r_ref, olens, *_ = reg(s_ref, ylens=ylens, n_quantizers=3, f0=None)    # infer_v2_5.py:835-838
        r_o, _ = C.length_regulator(rsd, rc, s_ref, ylens)
    print(f"codec.decode: ref {tuple(s_ref.shape)} rms {s_ref.pow(2).mean().sqrt():.3f} oracle max|d| = {(s_ref - s_o).abs().max():.3e}")
    print(f"length_regulator: ref {tuple(r_ref.shape)} rms {r_ref.pow(2).mean().sqrt():.3f} oracle max|d| = {(r_ref - r_o).abs().max():.3e}")
    np.savez_compressed(os.path.join(GOLD, f"{tag}.npz"), codes=codes.numpy(), s_infer=s_ref.numpy(), ylens=ylens.numpy(), cond=r_ref.numpy(),
                        seed=np.int64(seed),
                        codec_cfg=np.array([cc.codebook_size, cc.hidden_size, cc.codebook_dim, cc.vocos_dim, cc.vocos_intermediate_dim, cc.vocos_num_layers]),
                        reg_cfg=np.array([rc.channels, rc.in_channels, rc.n_layers, rc.groups, rc.codebook_size]))


if __name__ == "__main__":
    main()
    # widths the HIP engine's f32 GEMM / LayerNorm kernels take (multiples of 64 / 16): the fixture of tests/test_gpu_codec.py
    main("codec_e64", C.CodecConfig(codebook_size=96, hidden_size=64, codebook_dim=8, vocos_dim=64, vocos_intermediate_dim=128, vocos_num_layers=3),
         C.RegulatorConfig(channels=64, in_channels=64, n_layers=4, groups=1, codebook_size=64), seed=73, n_codes=21)
