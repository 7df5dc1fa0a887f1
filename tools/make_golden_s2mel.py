"""Mint tests/golden/s2mel_cfm.npz by running the REFERENCE's own CFM / DiT classes (indextts/s2mel/modules/flow_matching.py,
diffusion_transformer.py, gpt_fast/model.py, wavenet.py) on CPU with the seeded weights of oracle.s2mel_oracle.synth_weights,
and print the oracle-vs-reference agreement.  torchaudio / librosa / munch are stubbed (tools/ref_shim_s2mel.py): the
flow-matching path never calls them."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
import ref_shim_s2mel as R  # noqa: E402

R.install()
from indextts.s2mel.modules.flow_matching import CFM  # noqa: E402

from oracle import s2mel_oracle as S  # noqa: E402

GOLD = os.path.join(HERE, "..", "tests", "golden")


def reference(cfg: S.S2MelConfig, sd):
    args = R.munchify(dict(
        dit_type="DiT", reg_loss_type="l1",
        DiT=dict(hidden_dim=cfg.hidden_dim, num_heads=cfg.num_heads, depth=cfg.depth, class_dropout_prob=0.1, block_size=8192,
                 in_channels=cfg.in_channels, style_condition=True, final_layer_type="wavenet", target="mel",
                 content_dim=cfg.content_dim, content_codebook_size=cfg.content_codebook_size, content_type="discrete",
                 f0_condition=False, n_f0_bins=512, content_codebooks=1, is_causal=False, long_skip_connection=True,
                 zero_prompt_speech_token=False, time_as_token=False, style_as_token=False, uvit_skip_connection=True,
                 add_resblock_in_transformer=False),
        wavenet=dict(hidden_dim=cfg.wavenet_hidden, num_layers=cfg.wavenet_layers, kernel_size=cfg.wavenet_kernel,
                     dilation_rate=cfg.wavenet_dilation_rate, p_dropout=0.2, style_condition=True),
        style_encoder=dict(dim=cfg.style_dim)))
    m = CFM(args).eval()
    ref_keys = [k for k in m.state_dict().keys() if not k.endswith("input_pos")]
    ours = [n for n, _ in S.param_shapes(cfg)]
    assert ref_keys == ours, (set(ref_keys) ^ set(ours))
    for n, shp in S.param_shapes(cfg):
        assert tuple(m.state_dict()[n].shape) == shp, (n, shp, tuple(m.state_dict()[n].shape))
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("input_pos") for k in missing), (missing, unexpected)
    m.estimator.setup_caches(max_batch_size=2, max_seq_length=512)
    return m


def main():
    cfg = S.S2MelConfig(hidden_dim=64, num_heads=2, depth=5, in_channels=80, content_dim=48, style_dim=24, wavenet_hidden=64,
                        wavenet_layers=3, wavenet_kernel=5, wavenet_dilation_rate=2)
    seed = 61
    sd = S.synth_weights(cfg, seed)
    m = reference(cfg, sd)
    g = torch.Generator().manual_seed(seed + 1)
    T, T_prompt, n_steps, cfg_rate = 57, 19, 4, 0.7
    z = torch.randn(1, cfg.in_channels, T, generator=g)
    prompt = torch.randn(1, cfg.in_channels, T_prompt, generator=g) * 0.5 - 1.0
    mu = torch.randn(1, T, cfg.content_dim, generator=g)
    style = torch.randn(1, cfg.style_dim, generator=g)
    x_lens = torch.tensor([T - 6])                          # 6 padded frames: the key mask and the WaveNet mask bite
    with torch.no_grad():
        # one estimator call (CFG-stacked batch of 2) and the whole Euler solve, both from the reference classes
        t = torch.tensor([0.35, 0.35])
        px = torch.zeros(1, cfg.in_channels, T)
        px[..., :T_prompt] = prompt
        d_ref = m.estimator(torch.cat([z, z]), torch.cat([px, torch.zeros_like(px)]), x_lens, t,
                            torch.cat([style, torch.zeros_like(style)]), torch.cat([mu, torch.zeros_like(mu)]))
        d_o = S.dit_forward(sd, cfg, torch.cat([z, z]), torch.cat([px, torch.zeros_like(px)]), x_lens, t,
                            torch.cat([style, torch.zeros_like(style)]), torch.cat([mu, torch.zeros_like(mu)]))
        t_span = torch.linspace(0, 1, n_steps + 1)
        y_ref = m.solve_euler(z.clone(), x_lens, prompt, mu.clone(), style, None, t_span, inference_cfg_rate=cfg_rate)
        y_o = S.cfm_solve_euler(sd, cfg, z, x_lens, prompt, mu, style, n_steps, cfg_rate)
    print(f"estimator: ref {tuple(d_ref.shape)} rms {d_ref.pow(2).mean().sqrt():.3f}  oracle max|d| = {(d_ref - d_o).abs().max():.3e}")
    print(f"solve_euler ({n_steps} steps, cfg {cfg_rate}): ref rms {y_ref.pow(2).mean().sqrt():.3f}  oracle max|d| = {(y_ref - y_o).abs().max():.3e}")
    np.savez_compressed(os.path.join(GOLD, "s2mel_cfm.npz"), z=z.numpy(), prompt=prompt.numpy(), mu=mu.numpy(), style=style.numpy(),
                        x_lens=x_lens.numpy(), t=t.numpy(), estimator_out=d_ref.numpy(), euler_out=y_ref.numpy(),
                        n_steps=np.int64(n_steps), cfg_rate=np.float64(cfg_rate), seed=np.int64(seed),
                        cfg=np.array([cfg.hidden_dim, cfg.num_heads, cfg.depth, cfg.in_channels, cfg.content_dim, cfg.style_dim,
                                      cfg.wavenet_hidden, cfg.wavenet_layers, cfg.wavenet_kernel, cfg.wavenet_dilation_rate]))


def main_hd64():
    """Second fixture for the HIP engine (head_dim 64, all GEMM widths multiples of 64): two utterances of different length and
    prompt length, each run through the reference at batch 1 (its only mode, infer_v2_5.py:201) -- one estimator call and the
    CFG Euler solve per utterance.  The engine is tested per utterance and on the packed two-utterance batch."""
    cfg = S.S2MelConfig(hidden_dim=128, num_heads=2, depth=5, in_channels=80, content_dim=64, style_dim=32, wavenet_hidden=128,
                        wavenet_layers=3, wavenet_kernel=5, wavenet_dilation_rate=2)
    seed = 67
    sd = S.synth_weights(cfg, seed)
    m = reference(cfg, sd)
    g = torch.Generator().manual_seed(seed + 1)
    n_steps, cfg_rate = 4, 0.7
    out = {}
    for u, (T, Tp, pad) in enumerate(((57, 19, 6), (83, 30, 0))):
        z = torch.randn(1, cfg.in_channels, T, generator=g)
        prompt = torch.randn(1, cfg.in_channels, Tp, generator=g) * 0.5 - 1.0
        mu = torch.randn(1, T, cfg.content_dim, generator=g)
        style = torch.randn(1, cfg.style_dim, generator=g)
        x_lens = torch.tensor([T - pad])
        with torch.no_grad():
            t = torch.tensor([0.35, 0.35])
            px = torch.zeros(1, cfg.in_channels, T)
            px[..., :Tp] = prompt
            d_ref = m.estimator(torch.cat([z, z]), torch.cat([px, torch.zeros_like(px)]), x_lens, t,
                                torch.cat([style, torch.zeros_like(style)]), torch.cat([mu, torch.zeros_like(mu)]))
            d_o = S.dit_forward(sd, cfg, torch.cat([z, z]), torch.cat([px, torch.zeros_like(px)]), x_lens, t,
                                torch.cat([style, torch.zeros_like(style)]), torch.cat([mu, torch.zeros_like(mu)]))
            t_span = torch.linspace(0, 1, n_steps + 1)
            y_ref = m.solve_euler(z.clone(), x_lens, prompt, mu.clone(), style, None, t_span, inference_cfg_rate=cfg_rate)
            y_o = S.cfm_solve_euler(sd, cfg, z, x_lens, prompt, mu, style, n_steps, cfg_rate)
        print(f"hd64 utt {u} (T={T}, prompt {Tp}, x_lens {T - pad}): estimator rms {d_ref.pow(2).mean().sqrt():.3f} oracle max|d| "
              f"{(d_ref - d_o).abs().max():.3e}; solve_euler rms {y_ref.pow(2).mean().sqrt():.3f} oracle max|d| {(y_ref - y_o).abs().max():.3e}")
        out.update({f"z{u}": z.numpy(), f"prompt{u}": prompt.numpy(), f"mu{u}": mu.numpy(), f"style{u}": style.numpy(),
                    f"x_lens{u}": x_lens.numpy(), f"estimator_out{u}": d_ref.numpy(), f"euler_out{u}": y_ref.numpy()})
    np.savez_compressed(os.path.join(GOLD, "s2mel_cfm_hd64.npz"), t=np.float32(0.35), n_steps=np.int64(n_steps),
                        cfg_rate=np.float64(cfg_rate), seed=np.int64(seed), n_utts=np.int64(2),
                        cfg=np.array([cfg.hidden_dim, cfg.num_heads, cfg.depth, cfg.in_channels, cfg.content_dim, cfg.style_dim,
                                      cfg.wavenet_hidden, cfg.wavenet_layers, cfg.wavenet_kernel, cfg.wavenet_dilation_rate]), **out)


def prod_inputs(cfg, seed, T, Tp):
    """Inputs of the production-width fixture; `mu` (T x 512: the bulk) is regenerated from the seed by the tests instead of being stored --
    the fixture carries its sum as a drift check of the generator."""
    g = torch.Generator().manual_seed(seed + 1)
    z = torch.randn(1, cfg.in_channels, T, generator=g)
    prompt = torch.randn(1, cfg.in_channels, Tp, generator=g) * 0.5 - 1.0
    style = torch.randn(1, cfg.style_dim, generator=g)
    mu = torch.randn(1, T, cfg.content_dim, generator=torch.Generator().manual_seed(seed + 2))
    return z, prompt, mu, style


def main_prod():
    """Third fixture: the PRODUCTION widths (S2MelConfig defaults = the v2 / v2.5 checkpoint config: DiT 13 x 512 x 8 heads, WaveNet 8 x 512,
    content 512, style 192) and the production solve depth (25 CFG Euler steps), one utterance of 211 frames behind a 73-frame prompt with 9
    padded frames, run through the reference's own classes.  Pins the oracle (and, on the GPU, the engine's f32 mode) at the benchmarked
    architecture rather than at a miniature."""
    cfg = S.S2MelConfig()
    seed, T, Tp, pad, n_steps, cfg_rate = 71, 211, 73, 9, 25, 0.7
    sd = S.synth_weights(cfg, seed)
    m = reference(cfg, sd)
    z, prompt, mu, style = prod_inputs(cfg, seed, T, Tp)
    x_lens = torch.tensor([T - pad])
    with torch.no_grad():
        t = torch.tensor([0.35, 0.35])
        px = torch.zeros(1, cfg.in_channels, T)
        px[..., :Tp] = prompt
        d_ref = m.estimator(torch.cat([z, z]), torch.cat([px, torch.zeros_like(px)]), x_lens, t,
                            torch.cat([style, torch.zeros_like(style)]), torch.cat([mu, torch.zeros_like(mu)]))
        d_o = S.dit_forward(sd, cfg, torch.cat([z, z]), torch.cat([px, torch.zeros_like(px)]), x_lens, t,
                            torch.cat([style, torch.zeros_like(style)]), torch.cat([mu, torch.zeros_like(mu)]))
        y_ref = m.solve_euler(z.clone(), x_lens, prompt, mu.clone(), style, None, torch.linspace(0, 1, n_steps + 1), inference_cfg_rate=cfg_rate)
        y_o = S.cfm_solve_euler(sd, cfg, z, x_lens, prompt, mu, style, n_steps, cfg_rate)
    print(f"production widths (T={T}, prompt {Tp}, x_lens {T - pad}, {n_steps} steps): estimator rms {d_ref.pow(2).mean().sqrt():.3f} oracle max|d| "
          f"{(d_ref - d_o).abs().max():.3e}; solve_euler rms {y_ref.pow(2).mean().sqrt():.3f} oracle max|d| {(y_ref - y_o).abs().max():.3e}")
    np.savez_compressed(os.path.join(GOLD, "s2mel_cfm_prod.npz"), z=z.numpy(), prompt=prompt.numpy(), style=style.numpy(), x_lens=x_lens.numpy(),
                        mu_sum=np.float64(mu.double().sum()), t=np.float32(0.35), estimator_out=d_ref.numpy(), euler_out=y_ref.numpy(),
                        n_steps=np.int64(n_steps), cfg_rate=np.float64(cfg_rate), seed=np.int64(seed), T=np.int64(T), Tp=np.int64(Tp))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "prod":
        main_prod()
    else:
        main()
        main_hd64()
        main_prod()
