"""GPU parity tests of the s2mel flow-matching decoder (DiT estimator + CFG Euler solver) on the HIP engine, through the C ABI.

Checkers: `tests/golden/s2mel_cfm_hd64.npz` -- outputs of the REFERENCE's own `CFM` / `DiT` classes run at batch 1 on the
oracle's seeded weights (tools/make_golden_s2mel.py) -- and `oracle/s2mel_oracle.py` for pieces the fixture does not isolate.
Bars: f32 engine mode within 1e-4 (absolute, values of RMS ~1) of the reference outputs; bf16 mode (bf16 GEMM operands, Q/K/V
and probabilities; f32 accumulation, residual stream, norms, softmax statistics) within the bounds written below.
"""
import os

import numpy as np
import pytest
import torch

from oracle import s2mel_oracle as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F32_TOL = 1e-4
BF16_ESTIMATOR_RMS = 0.03          # RMS error of one estimator call (outputs of RMS ~1)
BF16_EULER_RMS = 0.03              # RMS error after the 4-step CFG solve


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "s2mel_cfm_hd64.npz"))
    c = [int(v) for v in z["cfg"]]
    cfg = S.S2MelConfig(hidden_dim=c[0], num_heads=c[1], depth=c[2], in_channels=c[3], content_dim=c[4], style_dim=c[5],
                        wavenet_hidden=c[6], wavenet_layers=c[7], wavenet_kernel=c[8], wavenet_dilation_rate=c[9])
    return z, cfg, S.synth_weights(cfg, int(z["seed"]))


def args_of(cfg):
    return dict(DiT=dict(hidden_dim=cfg.hidden_dim, num_heads=cfg.num_heads, depth=cfg.depth, in_channels=cfg.in_channels,
                         content_dim=cfg.content_dim, style_condition=True, final_layer_type="wavenet", is_causal=False,
                         long_skip_connection=True, uvit_skip_connection=True, time_as_token=False, style_as_token=False),
                wavenet=dict(hidden_dim=cfg.wavenet_hidden, num_layers=cfg.wavenet_layers, kernel_size=cfg.wavenet_kernel,
                             dilation_rate=cfg.wavenet_dilation_rate, style_condition=True),
                style_encoder=dict(dim=cfg.style_dim))


def engine(cfg, sd, precision):
    from indextts_amd import s2mel
    m = s2mel.CFM(args_of(cfg), precision=precision, device=DEV)
    m.load_state_dict(sd)
    return m


def rms(a):
    return float(torch.as_tensor(a).double().pow(2).mean().sqrt())


def utt(z, u):
    g = lambda k: torch.from_numpy(z[f"{k}{u}"])
    return g("z"), g("prompt"), g("mu"), g("style"), g("x_lens")


@pytest.mark.parametrize("prec", [0, 1])
def test_rope_attention_unit(prec):
    """RoPE + split + non-causal masked attention (the flash kernel in bf16 mode) vs torch, ragged sequences incl. one whose
    valid length is shorter than the frames processed and lengths that are not multiples of the 64-key tile."""
    import ctypes as C
    from indextts_amd import _lib
    heads, H = 2, 128
    frame = [70, 130, 5, 64]
    valid = [61, 130, 5, 64]
    g = torch.Generator().manual_seed(3)
    n_tok, t_max = sum(frame), max(frame)
    qkv = torch.randn(n_tok, 3 * H, generator=g)
    cfg = S.S2MelConfig(hidden_dim=H, num_heads=heads)
    tab = S.rope_table(cfg, t_max)
    seq_T = torch.tensor(frame, dtype=torch.int32)
    seq_len = torch.tensor(valid, dtype=torch.int32)
    seq_start = torch.cumsum(seq_T, 0, dtype=torch.int32) - seq_T
    tok_seq = torch.repeat_interleave(torch.arange(len(frame), dtype=torch.int32), seq_T.long())
    tok_t = torch.arange(n_tok, dtype=torch.int32) - seq_start[tok_seq.long()]
    # torch reference, per sequence
    ref = torch.zeros(n_tok, H)
    for s, (T, n) in enumerate(zip(frame, valid)):
        o = int(seq_start[s])
        q, k, v = qkv[o:o + T].split(H, dim=-1)
        q = S.apply_rope(q.view(1, T, heads, 64), tab[:T]).transpose(1, 2)
        k = S.apply_rope(k.view(1, T, heads, 64), tab[:T]).transpose(1, 2)
        v = v.view(1, T, heads, 64).transpose(1, 2)
        if prec == 1:
            q, k, v = (t.bfloat16().float() for t in (q, k, v))
        sc = (q @ k.transpose(-1, -2)) / 8.0
        sc[..., n:] = float("-inf")
        ref[o:o + T] = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(T, H)
    L = _lib.lib()
    d = lambda t: t.to(DEV).contiguous()
    out = torch.empty(n_tok, H, dtype=torch.bfloat16 if prec == 1 else torch.float32, device=DEV)
    scratch = torch.empty(L.itts_s2mel_attention_scratch_bytes(n_tok, len(frame), heads, t_max, prec), dtype=torch.uint8, device=DEV)
    keep = [d(qkv), d(tab), d(tok_seq), d(tok_t), d(seq_start), d(seq_T), d(seq_len)]
    _lib.check(L.itts_s2mel_attention_forward(*[_lib.ptr(t) for t in keep], len(frame), n_tok, t_max, heads, prec, _lib.ptr(out),
                                              _lib.ptr(scratch), scratch.numel(), _lib.stream_ptr(torch.device(DEV))), "attention")
    err = float((out.float().cpu() - ref).abs().max())
    print(f"attention unit (prec {prec}): max|d| = {err:.3e}")
    assert err < (2e-2 if prec == 1 else 2e-5)


def test_estimator_f32_vs_reference(golden_dir):
    z, cfg, sd = load(golden_dir)
    m = engine(cfg, sd, "fp32")
    for u in range(int(z["n_utts"])):
        x, prompt, mu, style, x_lens = utt(z, u)
        T, Tp = x.shape[-1], prompt.shape[-1]
        px = torch.zeros_like(x)
        px[..., :Tp] = prompt
        t = torch.full((2,), float(z["t"]))
        d = m.estimator(torch.cat([x, x]), torch.cat([px, torch.zeros_like(px)]), x_lens, t,
                        torch.cat([style, torch.zeros_like(style)]), torch.cat([mu, torch.zeros_like(mu)])).cpu()
        ref = torch.from_numpy(z[f"estimator_out{u}"])
        err = float((d - ref).abs().max())
        print(f"estimator f32 utt {u}: max|d| vs reference = {err:.3e} (rms of the output {rms(ref):.3f})")
        assert err <= F32_TOL


def test_solve_euler_f32_vs_reference_single_and_batched(golden_dir):
    """The 4-step CFG Euler solve: per utterance (the reference's batch-1 call) and both utterances packed into one call
    (new capability) -- every utterance of the batch equals its own batch-1 result and the reference output."""
    z, cfg, sd = load(golden_dir)
    m = engine(cfg, sd, "fp32")
    n_steps, rate = int(z["n_steps"]), float(z["cfg_rate"])
    t_span = torch.linspace(0, 1, n_steps + 1)
    singles = []
    for u in range(2):
        x, prompt, mu, style, x_lens = utt(z, u)
        y = m.solve_euler(x.clone(), x_lens, prompt, mu, style, None, t_span, rate).cpu()
        ref = torch.from_numpy(z[f"euler_out{u}"])
        err = float((y - ref).abs().max())
        print(f"solve_euler f32 utt {u}: max|d| vs reference = {err:.3e}")
        assert err <= F32_TOL
        assert float(y[..., : prompt.shape[-1]].abs().max()) == 0.0          # prompt frames held at zero
        singles.append(y)
    # packed batch of the two utterances
    (x0, p0, mu0, s0, l0), (x1, p1, mu1, s1, l1) = utt(z, 0), utt(z, 1)
    T0, T1 = x0.shape[-1], x1.shape[-1]
    Tm = max(T0, T1)
    x = torch.zeros(2, cfg.in_channels, Tm)
    x[0, :, :T0], x[1, :, :T1] = x0[0], x1[0]
    mu = torch.zeros(2, Tm, cfg.content_dim)
    mu[0, :T0], mu[1, :T1] = mu0[0], mu1[0]
    Pm = max(p0.shape[-1], p1.shape[-1])
    prompt = torch.zeros(2, cfg.in_channels, Pm)
    prompt[0, :, : p0.shape[-1]], prompt[1, :, : p1.shape[-1]] = p0[0], p1[0]
    y = m.solve_euler(x, torch.cat([l0, l1]), prompt, mu, torch.cat([s0, s1]), None, t_span, rate,
                      prompt_lens=[p0.shape[-1], p1.shape[-1]], frame_lens=[T0, T1]).cpu()
    for u, T in ((0, T0), (1, T1)):
        err = float((y[u:u + 1, :, :T] - singles[u]).abs().max())
        print(f"solve_euler f32 utt {u} inside the packed batch vs alone: max|d| = {err:.3e}")
        assert err <= 1e-5
        assert float((y[u:u + 1, :, :T] - torch.from_numpy(z[f"euler_out{u}"])).abs().max()) <= F32_TOL


@pytest.mark.parametrize("mode", ["fp32", "fp32x3", "fp32x3-6"])
def test_production_widths_25_steps_f32_vs_reference(golden_dir, mode):
    """The benchmarked architecture (DiT 13 x 512 x 8 heads, WaveNet 8 x 512) and solve depth (25 CFG Euler steps) against outputs of the
    REFERENCE's own classes (tests/golden/s2mel_cfm_prod.npz, tools/make_golden_s2mel.py::main_prod): one estimator call and the whole solve,
    211 frames behind a 73-frame prompt, 9 padded frames -- in the native f32 mode, in the fp32x3 mode that carries the benchmark's headline (every
    GEMM operand as three bf16 planes, 8 plane products: VERDICT r3's condition for the ruling) and in its 6-product form.  Bound: 1e-4 absolute
    on mel values of RMS 1.4 (north_star's bar; measured 2e-5 in the f32 mode by the round-3 driver run)."""
    from indextts_amd import _lib
    from tests.test_oracle_s2mel import load_prod
    z, cfg, sd, mu = load_prod(golden_dir)
    with _lib.option_scope(x3_products=6 if mode.endswith("-6") else 8):
        _production_case(z, cfg, sd, mu, mode.split("-")[0], mode)


def _production_case(z, cfg, sd, mu, precision, mode):
    m = engine(cfg, sd, precision)
    x, prompt, style, x_lens = (torch.from_numpy(z[k]) for k in ("z", "prompt", "style", "x_lens"))
    Tp = prompt.shape[-1]
    px = torch.zeros_like(x)
    px[..., :Tp] = prompt
    d = m.estimator(torch.cat([x, x]), torch.cat([px, torch.zeros_like(px)]), x_lens, torch.full((2,), float(z["t"])),
                    torch.cat([style, torch.zeros_like(style)]), torch.cat([mu, torch.zeros_like(mu)])).cpu()
    e1 = float((d - torch.from_numpy(z["estimator_out"])).abs().max())
    y = m.solve_euler(x.clone(), x_lens, prompt, mu, style, None, torch.linspace(0, 1, int(z["n_steps"]) + 1), float(z["cfg_rate"])).cpu()
    e2 = float((y - torch.from_numpy(z["euler_out"])).abs().max())
    print(f"production widths vs the reference's classes [{mode}]: estimator max|d| {e1:.3e}, 25-step solve max|d| {e2:.3e} "
          f"(rms {rms(y - torch.from_numpy(z['euler_out'])):.3e}, output rms {rms(z['euler_out']):.3f})")
    assert e1 <= 1e-4 and e2 <= 1e-4
    assert float(y[..., :Tp].abs().max()) == 0.0


def test_bf16_mode_within_bounds(golden_dir):
    """The mode the benchmark runs: error of the estimator and of the full solve against the reference's fp32 outputs."""
    z, cfg, sd = load(golden_dir)
    m = engine(cfg, sd, "bf16")
    n_steps, rate = int(z["n_steps"]), float(z["cfg_rate"])
    t_span = torch.linspace(0, 1, n_steps + 1)
    for u in range(2):
        x, prompt, mu, style, x_lens = utt(z, u)
        T, Tp = x.shape[-1], prompt.shape[-1]
        n = int(x_lens[0])
        px = torch.zeros_like(x)
        px[..., :Tp] = prompt
        d = m.estimator(torch.cat([x, x]), torch.cat([px, torch.zeros_like(px)]), x_lens, torch.full((2,), float(z["t"])),
                        torch.cat([style, torch.zeros_like(style)]), torch.cat([mu, torch.zeros_like(mu)])).cpu()
        e1 = rms((d - torch.from_numpy(z[f"estimator_out{u}"]))[..., :n])
        y = m.solve_euler(x.clone(), x_lens, prompt, mu, style, None, t_span, rate).cpu()
        e2 = rms((y - torch.from_numpy(z[f"euler_out{u}"]))[..., :n])
        print(f"bf16 utt {u}: estimator rms error {e1:.4f} (bound {BF16_ESTIMATOR_RMS}), solve rms error {e2:.4f} (bound {BF16_EULER_RMS})")
        assert e1 <= BF16_ESTIMATOR_RMS and e2 <= BF16_EULER_RMS


def test_production_width_smoke():
    """The shipped widths (hidden 512, 8 heads, SwiGLU 1536, WaveNet 512 x 8, k = 5, dilation 1) at a few hundred frames:
    bf16 engine vs the CPU oracle run in fp32 on the same seeded weights (bound as above), two utterances packed."""
    cfg = S.S2MelConfig(depth=3, wavenet_layers=2, wavenet_dilation_rate=1)
    sd = S.synth_weights(cfg, 5)
    m = engine(cfg, sd, "bf16")
    g = torch.Generator().manual_seed(6)
    T, Tp = [150, 97], [40, 33]
    Tm = max(T)
    x = torch.randn(2, 80, Tm, generator=g)
    mu = torch.randn(2, Tm, cfg.content_dim, generator=g)
    prompt = torch.randn(2, 80, max(Tp), generator=g) * 0.5 - 1.0
    style = torch.randn(2, cfg.style_dim, generator=g)
    t_span = torch.linspace(0, 1, 3)
    y = m.solve_euler(x.clone(), torch.tensor(T), prompt, mu, style, None, t_span, 0.7, prompt_lens=Tp, frame_lens=T).cpu()
    for u in range(2):
        with torch.no_grad():
            ref = S.cfm_solve_euler(sd, cfg, x[u:u + 1, :, : T[u]], torch.tensor([T[u]]), prompt[u:u + 1, :, : Tp[u]],
                                    mu[u:u + 1, : T[u]], style[u:u + 1], 2, 0.7)
        e = rms(y[u:u + 1, :, : T[u]] - ref)
        print(f"production widths, utt {u}: bf16 engine vs fp32 oracle rms error {e:.4f} (output rms {rms(ref):.3f})")
        assert e <= BF16_EULER_RMS


@pytest.mark.parametrize("dil", [1, 2])
def test_tile_gemm_kernels_agree(dil):
    """The three bf16 tile GEMM kernels -- 128 x 128 (four waves), 256 x 256 (eight waves: counted-vmcnt LDS-DMA pipeline, two
    staggered wave groups) and 256 x 128 (four waves, three-stage DMA ring) -- issue the same MFMAs on the same fragments in the
    same K order and share their epilogues: the whole bf16 solve (fused wqkv + RoPE + transposed V^T store, SwiGLU, tap-mode
    WaveNet conv + gate, res/skip, residual, plain stores) must be BITWISE equal between them, on every one of three repeats (an
    LDS-DMA race would show up as an intermittent difference)."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "s2mel_probe.py")
    outs = []
    for v in ("0", "1", "2"):
        env = dict(os.environ, PROBE_OPTS=f"tile256={v}", PROBE_DIL=str(dil))
        r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1])
    assert float(outs[0].split()[2]) > 1e-3
    assert outs[0] == outs[1] == outs[2], outs


def test_f32_fast_path_vs_separate_kernels(tmp_path):
    """The f32 mode's fast path -- f32-MFMA tile GEMMs with the fused epilogues (RoPE + Q / K / V^T scatter, SwiGLU, tap-mode conv +
    gate, res/skip, shadows) and the f32-MFMA flash attention -- against the round-2 f32 path it replaces (register-path GEMM,
    separate element-wise kernels, one-wave-per-query scalar attention; the path the reference goldens pinned), same process setup:
      (a) switching ONLY the GEMM kernel (option f32_tile) leaves the whole solve bitwise unchanged (same MFMAs, same k order);
      (b) the full fast path agrees with the separate-kernel path to 2e-5 (libm gate functions in both; the flash softmax works in
          the exp2 domain and sums keys in a different order)."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "s2mel_probe.py")
    runs = {"fast": {}, "separate_tile": dict(PROBE_OPTS="s2mel_fused=0,f32_attn_scalar=1"),
            "separate_reg": dict(PROBE_OPTS="s2mel_fused=0,f32_attn_scalar=1,f32_tile=0")}
    digest, out = {}, {}
    for name, extra in runs.items():
        path = str(tmp_path / f"{name}.pt")
        env = dict(os.environ, PROBE_PREC="fp32", PROBE_DIL="2", PROBE_REPS="2", PROBE_SAVE=path, **extra)
        r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        digest[name] = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1]
        out[name] = torch.load(path)
    assert digest["separate_tile"] == digest["separate_reg"], digest
    d = float((out["fast"] - out["separate_reg"]).abs().max())
    print(f"f32 s2mel: fused f32-MFMA tile GEMMs + f32 flash attention vs the separate-kernel path: max|d| = {d:.3e} "
          f"(output rms {rms(out['separate_reg']):.3f})")
    assert rms(out["separate_reg"]) > 1e-3 and d <= 2e-5


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_dead_row_elimination_is_bit_identical(precision):
    """`solve_euler` runs the stages after the DiT's last attention (skip_linear, conv1, the WaveNet, the final layer, conv2) only on the
    tail of every sequence -- the frames from `prompt_len - receptive field` on: the Euler step never reads the estimator at prompt frames
    (flow_matching.py:107).  Every kept row must come out BIT-identical to the full computation (rows are independent in the GEMMs,
    the WaveNet's context is inside the halo), with ragged prompts incl. one shorter than the halo; and fewer GEMM FLOPs are issued."""
    cfg = S.S2MelConfig(depth=3, wavenet_layers=3, wavenet_dilation_rate=2)          # halo = 2 * (1 + 2 + 4) = 14 frames
    sd = S.synth_weights(cfg, 5)
    m = engine(cfg, sd, precision)
    g = torch.Generator().manual_seed(6)
    T, Tp = [391, 97, 258], [140, 33, 9]
    Tm = max(T)
    x = torch.randn(3, 80, Tm, generator=g)
    mu = torch.randn(3, Tm, cfg.content_dim, generator=g)
    prompt = torch.randn(3, 80, max(Tp), generator=g) * 0.5 - 1.0
    style = torch.randn(3, cfg.style_dim, generator=g)
    t_span = torch.linspace(0, 1, 4)
    out, flops = {}, {}
    m.set_profiling(True)
    for on in (False, True):
        m.prune_dead_rows = on
        out[on] = m.solve_euler(x.clone(), torch.tensor(T), prompt, mu, style, None, t_span, 0.7, prompt_lens=Tp, frame_lens=T).cpu()
        flops[on] = m.profile()["gemm"]["flops"]
    m.set_profiling(False)
    assert rms(out[False]) > 1e-3 and torch.equal(out[False], out[True])
    assert flops[True] < 0.97 * flops[False], flops
    print(f"{precision}: GEMM FLOPs of the solve {flops[False]:.3e} -> {flops[True]:.3e} with the prompt rows' dead tail work removed")


@pytest.mark.parametrize("mode", ["fp32", "fp32x3", "fp32x3:x3_waves=8", "fp32x3:x3_waves=4", "fp32x3:x3_products=8", "fp32x3:x3_sched=0",
                                  "fp32x3:x3_attn=0,x3_products=8,x3_sched=0", "bf16"])
def test_estimator_is_bit_stable_run_to_run_at_solve_size(mode):
    """The production architecture on two utterances of 517 + 1926 frames (9772 packed rows with the CFG branch: every tile GEMM launch runs
    several blocks per CU): three estimator calls and three one-step solves on the same inputs return the same BITS, with the cached workspace
    filled with NaN patterns in between (nothing may be read before it is written).  Round 4 found run-to-run differences in the bf16 mode and in
    the x3 GEMM variants that are not the default one when two of their blocks share a CU, and worked around them (two launches / one block per CU);
    round 5 traced them to the packed RoPE arithmetic of the fused wqkv epilogue (pf_rope4; profiles/r05a/capture.log, r05b/trace.log) -- every
    mode and variant, fused and at two blocks per CU, is held to determinism here."""
    import hashlib
    from indextts_amd import s2mel, synth
    from indextts_amd import _lib
    args = synth.S2MEL_V2
    prec, _, optstr = mode.partition(":")
    opts = {k: int(v) for k, v in (kv.split("=") for kv in optstr.split(",") if kv)}
    with _lib.option_scope(**opts):
        _bit_stable_body(prec, mode, args, hashlib, s2mel, synth)


def _bit_stable_body(prec, mode, args, hashlib, s2mel, synth):
    m = s2mel.CFM(args, precision=prec, device=DEV)
    m.load_state_dict(synth.s2mel_weights(args, seed=1234))
    g = torch.Generator().manual_seed(0)
    B, Tp, T = 2, 517, 517 + 1926
    x = torch.randn(B, 80, T, generator=g).to(DEV)
    mu = torch.randn(B, T, args["DiT"]["content_dim"], generator=g).to(DEV)
    prompt = (torch.randn(1, 80, Tp, generator=g) * 0.5 - 1.0).to(DEV)
    style = torch.randn(1, args["style_encoder"]["dim"], generator=g).to(DEV)
    lens = torch.full((B,), T)
    px = torch.zeros_like(x)
    px[..., :Tp] = prompt
    hsh = lambda y: hashlib.sha1(y.float().cpu().numpy().tobytes()).hexdigest()[:12]
    est, sol = [], []
    for _ in range(4):
        if m._ws is not None:
            m._ws.fill_(0xFF)
        est.append(hsh(m.estimator(torch.cat([x, x]), torch.cat([px, torch.zeros_like(px)]), lens, torch.full((2 * B,), 0.3),
                               torch.cat([style.expand(B, -1), torch.zeros(B, style.shape[1], device=DEV)]), torch.cat([mu, torch.zeros_like(mu)]))))
        m._ws.fill_(0xFF)
        y = m.solve_euler(x.clone(), lens, prompt, mu, style, None, torch.linspace(0, 1, 2), 0.7, frame_lens=[T] * B)
        assert bool(torch.isfinite(y).all())
        sol.append(hsh(y))
    print(f"{mode}: estimator bits {est}, one-step solve bits {sol}")
    assert len(set(est)) == 1 and len(set(sol)) == 1


def test_bf16_estimator_is_bit_stable_run_to_run_with_stage_trace():
    """The bf16 mode at production depth, 12 estimator calls on the same inputs with the engine's stage checksums on (itts_s2mel_set_trace: one
    order-independent 64-bit checksum per stage output, in launch order): every call gives the same checksums for every stage.  Round 4 found this
    mode differing run to run in about one call of two; the trace put the first differing stage in the Q / K tiles of the fused wqkv epilogue of
    the bf16 tile kernels (profiles/r04j), and round 5's stage captures showed one quarter-wave per differing call storing v2 c1 instead of
    v2 c1 - v3 s1 -- the SLP-packed RoPE arithmetic (pf_rope4 replaces it).  The mode runs the FUSED epilogue again.  The trace names the stage
    should a difference come back."""
    import collections
    from indextts_amd import _lib, s2mel, synth
    args = synth.S2MEL_V2
    m = s2mel.CFM(args, precision="bf16", device=DEV)
    m.load_state_dict(synth.s2mel_weights(args, seed=1234))
    g = torch.Generator().manual_seed(0)
    B, Tp, T = 2, 517, 517 + 1926
    x = torch.randn(B, 80, T, generator=g).to(DEV)
    mu = torch.randn(B, T, args["DiT"]["content_dim"], generator=g).to(DEV)
    prompt = (torch.randn(1, 80, Tp, generator=g) * 0.5 - 1.0).to(DEV)
    style = torch.randn(1, args["style_encoder"]["dim"], generator=g).to(DEV)
    lens = torch.full((B,), T)
    px = torch.zeros_like(x)
    px[..., :Tp] = prompt
    L = _lib.lib()
    cap = 1024
    buf = torch.zeros(cap, dtype=torch.int64, device=DEV)
    _lib.check(L.itts_s2mel_set_trace(m._h, _lib.ptr(buf), cap), "itts_s2mel_set_trace")
    runs = []
    try:
        for _ in range(12):
            buf.zero_()
            m.estimator(torch.cat([x, x]), torch.cat([px, torch.zeros_like(px)]), lens, torch.full((2 * B,), 0.3),
                        torch.cat([style.expand(B, -1), torch.zeros(B, style.shape[1], device=DEV)]), torch.cat([mu, torch.zeros_like(mu)]))
            torch.cuda.synchronize()
            runs.append(tuple(buf[: L.itts_s2mel_trace_count(m._h)].cpu().tolist()))
        labels = [(L.itts_s2mel_trace_label(m._h, i) or b"?").decode() for i in range(len(runs[0]))]
    finally:
        _lib.check(L.itts_s2mel_set_trace(m._h, None, 0), "itts_s2mel_set_trace")
    major, cnt = collections.Counter(runs).most_common(1)[0]
    firsts = [next((labels[i] for i in range(len(major)) if r[i] != major[i]), "length") for r in runs if r != major]
    print(f"bf16 estimator, {len(major)} stage checksums per call: {cnt} of {len(runs)} calls agree on all of them; first differing stages: {firsts}")
    assert len(major) > 100 and cnt == len(runs), firsts


def test_x3_eight_wave_gemms_leave_the_estimator_bitwise_unchanged():
    """fp32x3 with option x3_waves = 8 against 4: every fused epilogue (wqkv + RoPE with the K / V^T planes, SwiGLU, residual, the tap-mode WaveNet
    conv with its gate, res-skip) runs on the 8-wave kernel and the estimator's output is BITWISE the 4-wave path's, at the production architecture
    on a ragged pair of utterances."""
    from indextts_amd import _lib, s2mel, synth
    args = synth.S2MEL_V2
    g = torch.Generator().manual_seed(5)
    B, Tp, T = 2, 100, 100 + 411
    x = torch.randn(B, 80, T, generator=g).to(DEV)
    mu = torch.randn(B, T, args["DiT"]["content_dim"], generator=g).to(DEV)
    prompt = (torch.randn(1, 80, Tp, generator=g) * 0.5 - 1.0).to(DEV)
    style = torch.randn(1, args["style_encoder"]["dim"], generator=g).to(DEV)
    lens = torch.tensor([T, T - 37])
    px = torch.zeros_like(x)
    px[..., :Tp] = prompt
    outs = []
    for waves in (4, 8):
        with _lib.option_scope(x3_waves=waves):
            m = s2mel.CFM(args, precision="fp32x3", device=DEV)
            m.load_state_dict(synth.s2mel_weights(args, seed=1234))
            outs.append(m.estimator(torch.cat([x, x]), torch.cat([px, torch.zeros_like(px)]), lens, torch.full((2 * B,), 0.3),
                                    torch.cat([style.expand(B, -1), torch.zeros(B, style.shape[1], device=DEV)]), torch.cat([mu, torch.zeros_like(mu)]),
                                    frame_lens=[T, T - 37] * 2).cpu())
            del m
    assert rms(outs[0]) > 1e-3 and torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())


def test_x3_plane_operands_are_bitwise_the_in_register_split():
    """fp32x3, option x3_aplanes = 1: the adaptive-RMSNorm outputs leave the norm kernel as three bf16 planes in fragment order and the wqkv / w1|w3
    GEMMs stage plane tiles instead of splitting f32 tiles in registers ("carry the planes across kernels").  Same planes, same
    MFMAs in the same order: the estimator's output is BITWISE the default path's, at the production architecture on a ragged pair of utterances.
    (Measured: +0.8 % on the solve -- the matrix pipe's idle cycles are not the split's, profiles/r04q -- so the option stays off by default.)"""
    from indextts_amd import _lib, s2mel, synth
    args = synth.S2MEL_V2
    g = torch.Generator().manual_seed(3)
    B, Tp, T = 2, 100, 100 + 411
    x = torch.randn(B, 80, T, generator=g).to(DEV)
    mu = torch.randn(B, T, args["DiT"]["content_dim"], generator=g).to(DEV)
    prompt = (torch.randn(1, 80, Tp, generator=g) * 0.5 - 1.0).to(DEV)
    style = torch.randn(1, args["style_encoder"]["dim"], generator=g).to(DEV)
    lens = torch.tensor([T, T - 37])
    px = torch.zeros_like(x)
    px[..., :Tp] = prompt
    outs = []
    for ap in (0, 1):
        with _lib.option_scope(x3_aplanes=ap):
            m = s2mel.CFM(args, precision="fp32x3", device=DEV)
            m.load_state_dict(synth.s2mel_weights(args, seed=1234))
            outs.append(m.estimator(torch.cat([x, x]), torch.cat([px, torch.zeros_like(px)]), lens, torch.full((2 * B,), 0.3),
                                    torch.cat([style.expand(B, -1), torch.zeros(B, style.shape[1], device=DEV)]), torch.cat([mu, torch.zeros_like(mu)]),
                                    frame_lens=[T, T - 37] * 2).cpu())
            del m
    assert rms(outs[0]) > 1e-3 and torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
