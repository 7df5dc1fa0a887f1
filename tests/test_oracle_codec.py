"""Oracle for the second "next" row (EnhancedCodec.decode + InterpolateRegulator) vs the fixture minted
from the reference's own classes (tools/make_golden_codec.py).  These tests pin the oracle; the HIP path is tested in tests/test_gpu_codec.py."""
import os

import numpy as np
import torch

from oracle import codec_oracle as C


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "codec.npz"))
    a, b = [int(v) for v in z["codec_cfg"]], [int(v) for v in z["reg_cfg"]]
    cc = C.CodecConfig(codebook_size=a[0], hidden_size=a[1], codebook_dim=a[2], vocos_dim=a[3], vocos_intermediate_dim=a[4], vocos_num_layers=a[5])
    rc = C.RegulatorConfig(channels=b[0], in_channels=b[1], n_layers=b[2], groups=b[3], codebook_size=b[4])
    seed = int(z["seed"])
    return z, cc, rc, C.synth_codec_weights(cc, seed), C.synth_regulator_weights(rc, seed + 1)


def test_codec_decode_matches_reference(golden_dir):
    z, cc, rc, csd, rsd = load(golden_dir)
    with torch.no_grad():
        s = C.codec_decode(csd, cc, torch.from_numpy(z["codes"]))
    assert s.shape == (z["codes"].shape[0], 2 * z["codes"].shape[1], cc.hidden_size)
    np.testing.assert_allclose(s.numpy(), z["s_infer"], rtol=0, atol=1e-5)


def test_length_regulator_matches_reference(golden_dir):
    z, cc, rc, csd, rsd = load(golden_dir)
    ylens = torch.from_numpy(z["ylens"])
    with torch.no_grad():
        cond, olens = C.length_regulator(rsd, rc, torch.from_numpy(z["s_infer"]), ylens)
    np.testing.assert_allclose(cond.numpy(), z["cond"], rtol=0, atol=1e-5)
    assert olens.tolist() == ylens.tolist()
    assert float(cond[1, int(ylens[1]):].abs().max()) == 0.0              # rows are zeroed beyond their own target length


def test_quantize_equals_reference_class(golden_dir):
    """oracle `codec_quantize` vs tests/golden/codec_quantize.npz = the reference's own EnhancedCodec.quantize (tools/make_golden_codec_quantize.py):
    identical indices, quantized features to f32 rounding."""
    from tools.make_golden_codec_quantize import CFG, LENGTHS, SEED
    z = np.load(os.path.join(golden_dir, "codec_quantize.npz"))
    sd = C.synth_codec_weights(CFG, SEED)
    sd.update(C.synth_codec_encoder_weights(CFG, SEED + 1))
    for i, T in enumerate(LENGTHS):
        idx, q, margin = C.codec_quantize(sd, CFG, torch.from_numpy(z[f"x{i}"]))
        assert idx.shape == (2, (T - 1) // 2 + 1) and np.array_equal(idx.numpy(), z[f"idx{i}"])
        assert float((q - torch.from_numpy(z[f"q{i}"])).abs().max()) <= 2e-6 and float(margin.min()) > 0
    idx, q, _ = C.codec_quantize(sd, CFG, torch.from_numpy(z["x_b1"]))
    assert np.array_equal(idx.numpy(), z["idx_b1"]) and idx.shape == (1, 5) and float((q - torch.from_numpy(z["q_b1"])).abs().max()) <= 2e-6
