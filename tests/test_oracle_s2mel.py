"""Oracle for the first "next" row (SURVEY.md section 8f-1, s2mel CFM / DiT) vs the fixture minted from the reference's own
CFM / DiT classes (tools/make_golden_s2mel.py).  There is no HIP path for this row yet: these tests pin the oracle only."""
import os

import numpy as np
import torch

from oracle import s2mel_oracle as S


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "s2mel_cfm.npz"))
    c = [int(v) for v in z["cfg"]]
    cfg = S.S2MelConfig(hidden_dim=c[0], num_heads=c[1], depth=c[2], in_channels=c[3], content_dim=c[4], style_dim=c[5],
                        wavenet_hidden=c[6], wavenet_layers=c[7], wavenet_kernel=c[8], wavenet_dilation_rate=c[9])
    return z, cfg, S.synth_weights(cfg, int(z["seed"]))


def test_estimator_matches_reference(golden_dir):
    z, cfg, sd = load(golden_dir)
    x = torch.from_numpy(z["z"])
    T, Tp = x.shape[-1], z["prompt"].shape[-1]
    px = torch.zeros_like(x)
    px[..., :Tp] = torch.from_numpy(z["prompt"])
    style, mu = torch.from_numpy(z["style"]), torch.from_numpy(z["mu"])
    with torch.no_grad():
        d = S.dit_forward(sd, cfg, torch.cat([x, x]), torch.cat([px, torch.zeros_like(px)]), torch.from_numpy(z["x_lens"]),
                          torch.from_numpy(z["t"]), torch.cat([style, torch.zeros_like(style)]), torch.cat([mu, torch.zeros_like(mu)]))
    assert d.shape == (2, cfg.in_channels, T)
    np.testing.assert_allclose(d.numpy(), z["estimator_out"], rtol=0, atol=2e-5)


def test_euler_solver_matches_reference(golden_dir):
    z, cfg, sd = load(golden_dir)
    with torch.no_grad():
        y = S.cfm_solve_euler(sd, cfg, torch.from_numpy(z["z"]), torch.from_numpy(z["x_lens"]), torch.from_numpy(z["prompt"]),
                              torch.from_numpy(z["mu"]), torch.from_numpy(z["style"]), int(z["n_steps"]), float(z["cfg_rate"]))
    np.testing.assert_allclose(y.numpy(), z["euler_out"], rtol=0, atol=2e-5)
    Tp = z["prompt"].shape[-1]
    assert float(y[..., :Tp].abs().max()) == 0.0                      # prompt frames are held at zero (flow_matching.py:112)


def test_padded_keys_do_not_reach_valid_frames(golden_dir):
    """sequence_mask semantics: frames beyond x_lens are never attended to; the transformer and WaveNet masks together make
    the valid frames independent of what the padded input frames hold except through the (unmasked) reflect-padded convs'
    3 * 4-frame reach at the boundary."""
    z, cfg, sd = load(golden_dir)
    x = torch.from_numpy(z["z"]).clone()
    T = x.shape[-1]
    n = int(z["x_lens"][0])
    px = torch.zeros_like(x)
    style, mu = torch.from_numpy(z["style"]), torch.from_numpy(z["mu"])
    t = torch.tensor([0.5])
    with torch.no_grad():
        a = S.dit_forward(sd, cfg, x, px, torch.tensor([n]), t, style, mu)
        x2 = x.clone()
        x2[..., n:] += 3.0
        b = S.dit_forward(sd, cfg, x2, px, torch.tensor([n]), t, style, mu)
    reach = sum((cfg.wavenet_kernel - 1) // 2 * cfg.wavenet_dilation_rate ** i for i in range(cfg.wavenet_layers)) + 1
    assert float((a - b)[..., : n - reach].abs().max()) < 1e-4


def load_prod(golden_dir):
    """Third fixture (tools/make_golden_s2mel.py::main_prod): the PRODUCTION widths (DiT 13 x 512 x 8 heads, WaveNet 8 x 512) and solve depth (25
    CFG Euler steps) run through the reference's classes; `mu` is regenerated from the seed (its sum is the fixture's drift check)."""
    z = np.load(os.path.join(golden_dir, "s2mel_cfm_prod.npz"))
    cfg = S.S2MelConfig()
    seed, T = int(z["seed"]), int(z["T"])
    mu = torch.randn(1, T, cfg.content_dim, generator=torch.Generator().manual_seed(seed + 2))
    assert abs(float(mu.double().sum()) - float(z["mu_sum"])) < 1e-6, "torch's CPU generator no longer reproduces the fixture's mu"
    return z, cfg, S.synth_weights(cfg, seed), mu


def test_production_widths_estimator_and_25_step_solve_match_reference(golden_dir):
    z, cfg, sd, mu = load_prod(golden_dir)
    x, prompt, style, x_lens = (torch.from_numpy(z[k]) for k in ("z", "prompt", "style", "x_lens"))
    Tp = prompt.shape[-1]
    px = torch.zeros_like(x)
    px[..., :Tp] = prompt
    with torch.no_grad():
        d = S.dit_forward(sd, cfg, torch.cat([x, x]), torch.cat([px, torch.zeros_like(px)]), x_lens, torch.full((2,), float(z["t"])),
                          torch.cat([style, torch.zeros_like(style)]), torch.cat([mu, torch.zeros_like(mu)]))
        y = S.cfm_solve_euler(sd, cfg, x, x_lens, prompt, mu, style, int(z["n_steps"]), float(z["cfg_rate"]))
    np.testing.assert_allclose(d.numpy(), z["estimator_out"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(y.numpy(), z["euler_out"], rtol=0, atol=3e-5)
    assert float(y[..., :Tp].abs().max()) == 0.0
