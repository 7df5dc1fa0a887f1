#!/usr/bin/env python3
"""Headline benchmark: audio-seconds generated per wall-second, IndexTTS-2.5 hot path, 64-utterance x 128-token batch.

One "step" = one pass of the hot path over the synthetic 64-utterance batch (BASELINE.json configs[2]):
    speaker-bundle broadcast (RCCL when N > 1)
    -> GPT speech-token decode of this rank's utterances (prefill + 560 sampled tokens each, top-k 30 / top-p 0.8 / T 0.8 /
       repetition penalty 10, bf16 weights + KV)
    -> codes -> mel: semantic-codec decode, length regulator, 25-step classifier-free-guidance flow matching over
       [speaker prompt | int(2 * n_tokens * 1.72) target frames] (indextts/infer_v2_5.py:830-846), bf16 GEMMs / attention
       (`--no-s2mel` vocodes a synthetic mel of that length instead: the round-1 hot-path-only measurement)
    -> BigVGAN (80-band mel -> 22.05 kHz wave, fp32) -> int16 -> waveforms gathered on rank 0.
Scaling is STRONG by default: the 64 utterances are LPT-sharded over the N ranks (`indextts_amd.dist.shard_utterances`),
64/N per GPU, as BASELINE.json's metric ("64-utt batch @1/2/4/8") says; `--weak` keeps 64 utterances per GPU instead.
Inputs (text ids, conditioning vectors, mel) are resident in HBM before the timed region.  Random-init weights of the
reference architecture (no checkpoints exist offline); EOS is suppressed so every row decodes all 560 tokens.

Contract: `python bench.py --gpus N --steps K --warmup W`; with N > 1 and no torchrun environment the script re-executes
itself under `python -m torch.distributed.run --nproc-per-node N` (one rank per GPU); rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32 MFMA = f32 vector peak
PEAK_HBM_GBPS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA (no sparsity)
SR, HOP = 22050, 256
EULER_STEPS = 25                  # diffusion steps of cfm.inference in the pipeline (indextts/infer_v2_5.py:843)
METRIC = "audio-seconds/sec (RTF) IndexTTS-2.5, 64-utt batch @1/2/4/8 MI355X"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask and cgroup CPU quota, not the host's os.cpu_count()
    (spawning one OpenMP thread per HOST core inside a quota-limited container stalls for minutes)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(gpt_sd, gpt_cfg, bv_sd, bv_h, n_text, n_gen, t_mel, threads, s2mel_frames=None):
    """Reference CPU path restated (oracle/), timed on this box's host cores on a bounded sample.

    GPT: 1 utterance, `n_text` text tokens, prefill + 120 greedy decode steps (fp32, kv-cache) on the full-size stack.
    BigVGAN: 1 utterance x 480 mel frames (fp32).  s2mel: one of the 25 CFG Euler steps at the bench's frame count.
    About 20-35 s of CPU work; scaled to audio-seconds/second for an
    utterance of `n_gen` tokens / `t_mel` frames (decode cost per token grows with context; the sample covers the first
    120 of `n_gen` positions, which favours the CPU).  The reference itself cannot run on the GPU box (/root/reference
    does not travel), hence kind = "port": the oracle is the restatement pinned to it by the committed fixtures.
    """
    from oracle import bigvgan_oracle as BO
    from oracle import gpt_oracle as GO
    cores = max(1, int(threads))
    torch.set_num_threads(cores)
    log(f"[bench] cpu_baseline on {cores} threads (os.cpu_count()={os.cpu_count()}, usable={usable_cores()})")
    cfg = GO.GPTConfig(layers=gpt_cfg["layers"], model_dim=gpt_cfg["model_dim"], heads=gpt_cfg["heads"],
                       max_text_tokens=gpt_cfg["max_text_tokens"], max_mel_tokens=gpt_cfg["max_mel_tokens"],
                       number_text_tokens=gpt_cfg["number_text_tokens"])
    g = torch.Generator().manual_seed(7)
    text = torch.randint(2, cfg.number_text_tokens, (1, n_text), generator=g)
    conds = torch.randn(1, 3, cfg.model_dim, generator=g) * 0.1
    steps = 120
    with torch.no_grad():
        fake, embeds, mask = GO.prepare_gpt_inputs(gpt_sd, cfg, conds, text, torch.tensor([3]))
        model = GO.InferenceModel(gpt_sd, cfg, kv_cache=True)
        model.store_mel_emb(embeds)
        t0 = time.perf_counter()
        logits, past = model.forward(fake, mask, None)
        t_prefill = time.perf_counter() - t0
        ids = fake
        t0 = time.perf_counter()
        for _ in range(steps):
            nxt = logits[:, -1].argmax(-1)
            ids = torch.cat([ids, nxt[:, None]], 1)
            mask = torch.cat([mask, mask.new_ones(1, 1)], 1)
            logits, past = model.forward(ids, mask, past)
        t_tok = (time.perf_counter() - t0) / steps
        frames = 480
        mel = torch.randn(1, bv_h["num_mels"], frames, generator=g) * 2 - 4
        BO.bigvgan_forward(bv_sd, mel[:, :, :8], bv_h)      # warm
        t0 = time.perf_counter()
        BO.bigvgan_forward(bv_sd, mel, bv_h)
        t_frame = (time.perf_counter() - t0) / frames
        # s2mel: ONE Euler step (one CFG-stacked estimator call of the DiT 13 x 512 + WaveNet 8 x 512) at the bench's frame count,
        # x euler_steps; skipped (and left out of `value`) when the GPU line runs without the s2mel stage
        t_euler = None
        if s2mel_frames:
            from oracle import s2mel_oracle as SO
            scfg = SO.S2MelConfig()
            ssd = SO.synth_weights(scfg, 3)
            T, Tp = s2mel_frames
            z = torch.randn(1, scfg.in_channels, T, generator=g)
            t0 = time.perf_counter()
            SO.cfm_solve_euler(ssd, scfg, z, torch.tensor([T]), torch.randn(1, scfg.in_channels, Tp, generator=g),
                               torch.randn(1, T, scfg.content_dim, generator=g), torch.randn(1, scfg.style_dim, generator=g), 1, 0.7)
            t_euler = time.perf_counter() - t0
    audio_s = t_mel * HOP / SR
    cpu_time = t_prefill + n_gen * t_tok + t_mel * t_frame + (EULER_STEPS * t_euler if t_euler is not None else 0.0)
    res = {"value": audio_s / cpu_time, "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
           "sample": f"oracle (torch fp32 CPU restatement of the reference path): GPT 1 utt x {n_text} text tokens, prefill + "
                     f"{steps} greedy decode steps; BigVGAN 1 utt x {frames} mel frames"
                     + (f"; s2mel 1 utt x {s2mel_frames[0]} frames, 1 of {EULER_STEPS} CFG Euler steps" if t_euler is not None else "")
                     + f"; extrapolated to {n_gen} tokens / {t_mel} frames / {EULER_STEPS} steps (codec decode + length regulator, "
                       "1 % of the GPU step, not included)",
           "gpt_ms_per_token": t_tok * 1e3, "gpt_prefill_ms": t_prefill * 1e3, "bigvgan_ms_per_frame": t_frame * 1e3}
    if t_euler is not None:
        res["s2mel_ms_per_euler_step"] = t_euler * 1e3
    return res


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no torchrun environment re-executes itself with one rank per GPU
# ---------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n: int) -> int:
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] launching", " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


# ---------------------------------------------------------------------------------------------------------------------
# engines
# ---------------------------------------------------------------------------------------------------------------------
class HipEngine:
    """The product path: indextts_amd host classes over libindextts_hip.so."""
    name = "hip"

    def __init__(self, args, dev, rank):
        from indextts_amd import bigvgan, gpt, synth
        t_load = time.perf_counter()
        self.args, self.dev = args, dev
        self.gcfg = dict(synth.GPT_V25)
        self.gsd = synth.gpt_weights(self.gcfg, seed=1234, suppress_eos=True)
        self.model = gpt.UnifiedVoice(**self.gcfg, spk_cond_mode="campplus", precision=args.precision, device=str(dev))
        self.model.load_state_dict(self.gsd)
        self.model.post_init_gpt2_config(kv_cache=True, half=args.precision == "bf16")
        self.model.use_graph = not args.no_graph
        self.bh = dict(synth.BIGVGAN_V2_22K)
        self.bsd = synth.bigvgan_weights(self.bh, seed=1234)
        self.voc = bigvgan.BigVGAN(self.bh, device=dev)
        self.voc.load_state_dict(self.bsd)
        self.voc.to(dev)
        self.voc.set_profiling(True)
        self.prof_acc = {}
        self.gpt_t = {"prefill_ms": 0.0, "decode_ms": 0.0, "steps": 0}
        self.s2 = None
        self.s2_t = {"codec_regulator_ms": 0.0, "cfm_ms": 0.0, "gemm_ms": 0.0, "gemm_flops": 0.0, "attention_ms": 0.0,
                     "attention_flops": 0.0, "estimator_ms": 0.0, "launches": 0}
        if not args.no_s2mel:
            from indextts_amd import codec, s2mel
            self.codec = codec.EnhancedCodec(**synth.CODEC_V2, device=dev)
            self.codec.load_state_dict(synth.codec_weights(seed=1234))
            self.s2_args = dict(synth.S2MEL_V2, length_regulator=synth.REGULATOR_V2)
            self.s2 = s2mel.MyModel(self.s2_args, precision=args.precision, device=dev)
            self.s2.models["cfm"].load_state_dict(synth.s2mel_weights(seed=1234))
            self.s2.models["length_regulator"].load_state_dict(synth.regulator_weights(seed=1234))
            self.s2.models["cfm"].set_profiling(True)
            self._ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        self.gen_kw = dict(do_sample=True, top_p=0.8, top_k=30, temperature=0.8, num_beams=1, repetition_penalty=10.0,
                           length_penalty=0.0)
        if rank == 0:
            log(f"[bench] weights synthesised + packed + uploaded in {time.perf_counter() - t_load:.1f}s")

    def step(self, text, langs, mel, bundle, n_gen, record):
        """-> int16 waveforms (b, T*256) of this rank's utterances"""
        return self.render(self.decode(text, langs, bundle, n_gen, record), mel, bundle, n_gen, record)

    def decode(self, text, langs, bundle, n_gen, record):
        """GPT stage: text ids -> speech codes (b, n_gen), on torch's CURRENT stream (host-blocking: the device loop reports back
        every 8 tokens)"""
        codes, _ = self.model.inference_speech(None, text, langs=langs, emo_vec=bundle["emo_vec"], campplus_embedding=bundle["style"],
                                               max_generate_length=n_gen, **self.gen_kw)
        assert codes.shape == (text.shape[0], n_gen), codes.shape
        if record:
            for k in ("prefill_ms", "decode_ms", "steps"):
                self.gpt_t[k] += self.model.last_timing[k]
        return codes

    def render(self, codes, mel, bundle, n_gen, record):
        """codes -> codec decode -> length regulator -> 25-step CFG flow matching -> BigVGAN -> int16, on torch's current stream"""
        B = codes.shape[0]
        style = bundle["style"]
        if self.s2 is not None:
            # codes -> content features -> 25-step CFG flow matching -> mel (indextts/infer_v2_5.py:830-846), all on the engine
            from indextts_amd import s2mel
            cfm = self.s2.models["cfm"]
            self._ev[0].record()
            S_infer = self.codec.decode(codes, code_lens=[n_gen] * B)
            target = [int(2 * n_gen * 1.72)] * B
            cond = self.s2.models["length_regulator"](S_infer, ylens=torch.tensor(target), n_quantizers=3, f0=None,
                                                      xlens=[2 * n_gen] * B, frame_lens=target)[0]
            self._ev[1].record()
            Tp = int(bundle["prompt_condition"].shape[1])
            cat = torch.cat([bundle["prompt_condition"].expand(B, -1, -1), cond], dim=1)
            total = [Tp + target[0]] * B
            mel = cfm.inference(cat, torch.tensor(total), bundle["ref_mel"], style, None, 25, inference_cfg_rate=0.7,
                                frame_lens=total)[:, :, Tp:].contiguous()
            self._ev[2].record()
            assert mel.shape == (B, 80, target[0]), mel.shape
            if record:
                pr = cfm.profile()                                 # synchronises the launch stream
                t = self.s2_t
                t["codec_regulator_ms"] += self._ev[0].elapsed_time(self._ev[1])
                t["cfm_ms"] += self._ev[1].elapsed_time(self._ev[2])
                t["gemm_ms"] += pr["gemm"]["ms"]
                t["gemm_flops"] += pr["gemm"]["flops"]
                t["attention_ms"] += pr["attention"]["ms"]
                t["attention_flops"] += 4.0 * cfm.hidden_dim * (2 * B) * float(total[0]) ** 2 * pr["attention"]["launches"]
                t["estimator_ms"] += pr["estimator_calls"]["ms"]
                t["launches"] += pr["gemm"]["launches"] + pr["attention"]["launches"]
        chunk = self.args.bigvgan_chunk or B
        outs = []
        for b0 in range(0, B, chunk):
            outs.append(self.voc(mel[b0:b0 + chunk]))
            if record:
                for k, v in self.voc.profile().items():
                    a = self.prof_acc.setdefault(k, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
                    for kk in a:
                        a[kk] += v[kk]
        wav = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
        return torch.clamp(32767.0 * wav[:, 0], -32767.0, 32767.0).to(torch.int16)      # infer_v2_5.py:855, :897-898

    # short untimed extras (rank 0, N = 1): the cost of the other GPT modes at the bench shape
    def extra_modes(self, text, langs, style, emo_vec, n_tok=32, t_mel=None):
        from indextts_amd import bigvgan, gpt
        out = {}
        if t_mel:
            # the opt-in second vocoder mode: resblock convs with f32 operands split into two f16 parts, three
            # exact f16 MFMA products per f32 product.  One untimed-by-the-headline forward at the bench shape, beside the f32 mode.
            B = text.shape[0]
            mel = torch.randn(B, 80, t_mel, generator=torch.Generator().manual_seed(7)).to(self.dev) * 2 - 4
            v3 = bigvgan.BigVGAN(self.bh, device=self.dev, conv_mode="f16x3")
            v3.load_state_dict(self.bsd)
            v3.to(self.dev)
            v3.set_profiling(True)
            w3 = v3(mel)
            w3 = v3(mel)
            pr = v3.profile()
            ref = self.voc(mel)
            self.voc.profile()
            conv = pr["conv1d_mfma"]
            tf = conv["flops"] / (conv["ms"] * 1e-3) / 1e12
            out["bigvgan_f16x3_mode"] = {
                "dtype": "resblock convs: f32 operands as two f16 parts each (22 significand bits), 3 exact f16 MFMA products per f32 "
                         "product, f32 accumulate; everything else f32",
                "ms_per_step": sum(v["ms"] for v in pr.values()),
                "conv_ms_per_step": conv["ms"], "conv_tflops_f32_equivalent": tf,
                "roofline": {"bound": "mfma", "kernel": "conv_h3_kernel + split pass (3 x v_mfma_f32_16x16x32_f16 per product) and the "
                             "f32 kernel on the narrow stages", "achieved": tf, "peak": PEAK_BF16_MFMA_TFLOPS / 3.0,
                             "unit": "TFLOP/s (f32-equivalent)", "frac": tf / (PEAK_BF16_MFMA_TFLOPS / 3.0)},
                "rms_vs_f32_mode": float((w3.float() - ref.float()).pow(2).mean().sqrt()),
                "signal_rms": float(ref.float().pow(2).mean().sqrt())}
            del v3, w3, ref
            torch.cuda.empty_cache()
        kw = dict(self.gen_kw, num_beams=3)                    # the reference default: 3-beam beam-sample
        for _ in range(2):
            self.model.inference_speech(None, text, langs=langs, emo_vec=emo_vec, campplus_embedding=style,
                                        max_generate_length=n_tok, **kw)
        t = self.model.last_timing
        out["gpt_beam3_sample_ms_per_token"] = t["decode_ms"] / max(1, t["steps"] - 1)
        if self.args.precision == "bf16":
            m32 = gpt.UnifiedVoice(**self.gcfg, spk_cond_mode="campplus", precision="fp32", device=str(self.dev))
            m32.load_state_dict(self.gsd)
            m32.post_init_gpt2_config(kv_cache=True)
            for _ in range(2):
                m32.inference_speech(None, text, langs=langs, emo_vec=emo_vec, campplus_embedding=style,
                                     max_generate_length=n_tok, **self.gen_kw)
            t = m32.last_timing
            out["gpt_f32_mode_ms_per_token"] = t["decode_ms"] / max(1, t["steps"] - 1)     # the ids-bit-exact parity mode
            out["gpt_f32_mode_prefill_ms"] = t["prefill_ms"]
            del m32
            torch.cuda.empty_cache()
        return out


class StubEngine:
    """CPU stand-in used ONLY by tests/test_bench_launcher.py (`--engine stub`, gloo): exercises this file's launcher,
    sharding, broadcast, gather and timing protocol without a GPU.  Its output is labelled as a stub and is not a result."""
    name = "stub"

    def __init__(self, args, dev, rank):
        self.prof_acc, self.gpt_t = {}, {"prefill_ms": 0.0, "decode_ms": 0.0, "steps": 0}

    def step(self, text, langs, mel, bundle, n_gen, record):
        return self.render(self.decode(text, langs, bundle, n_gen, record), mel, bundle, n_gen, record)

    def decode(self, text, langs, bundle, n_gen, record):
        return (text[:, :1].to(torch.int64) % 97).to(torch.int16) + int(bundle["style"].double().sum() * 0)   # identifies the utterance

    def render(self, codes, mel, bundle, n_gen, record):
        return codes.expand(codes.shape[0], mel.shape[-1] * HOP).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utts", type=int, default=64, help="utterances in the batch (strong scaling: sharded over the ranks; "
                                                          "--weak: per GPU)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --utts utterances PER GPU")
    ap.add_argument("--text-tokens", type=int, default=128)
    ap.add_argument("--gen-tokens", type=int, default=560)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--bigvgan-chunk", type=int, default=0, help="utterances per BigVGAN launch group (0 = all)")
    ap.add_argument("--no-s2mel", action="store_true", help="skip codes -> mel (codec, length regulator, 25-step CFM) and vocode a "
                                                           "synthetic mel instead (the round-1 hot-path-only measurement)")
    ap.add_argument("--prompt-frames", type=int, default=517, help="reference-speaker prompt length in mel frames (6 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline leg (0 = all usable cores)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed beam-3 / f32-mode decode measurements")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--overlap", action="store_true", help="software-pipeline consecutive steps: the GPT decode of step k + 1 (latency-bound, "
                    "small grids) runs on a second, high-priority HIP stream from its own host thread while step k's codec / flow "
                    "matching / vocoder kernels (compute-bound) run on the main stream; every step still completes inside the timed region")
    ap.add_argument("--engine", default="hip", choices=["hip", "stub"], help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    # stdout carries exactly ONE line (the JSON of rank 0): native libraries (RCCL prints a version banner on init) and any
    # stray print write to fd 1, so fd 1 is pointed at stderr for the run and the result goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus})")
    stub = args.engine == "stub"
    if stub:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("ITTS_BENCH_FORCE_DIST") == "1":      # the env switch exercises the RCCL path on one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    from indextts_amd import dist as D

    eng = (StubEngine if stub else HipEngine)(args, dev, rank)
    n_text, n_gen = args.text_tokens, args.gen_tokens
    t_mel = int(2 * n_gen * 1.72)
    n_total = args.utts * (world if args.weak else 1)
    # the whole synthetic batch is generated identically on every rank (seeded), then sharded: strong scaling hands each
    # rank its LPT share of the SAME 64 utterances (all 128 text tokens long here, so the shares are equal)
    g = torch.Generator().manual_seed(100)
    n_text_ids, n_mels, D_model = (12000, 80, 1280) if stub else (eng.gcfg["number_text_tokens"], eng.bh["num_mels"], eng.gcfg["model_dim"])
    text_all = torch.randint(2, n_text_ids, (n_total, n_text), generator=g)
    mine = D.shard_utterances(n_total, rank, world, lengths=[n_text] * n_total)
    B = len(mine)
    text = text_all[mine].to(dev)
    langs = torch.full((B,), 3, dtype=torch.long, device=dev)
    mel = (torch.randn(B, n_mels, t_mel, generator=torch.Generator().manual_seed(200 + rank)) * 2 - 4).to(dev)
    # speaker bundle: produced by the prompt encoders on rank 0 in the real pipeline, broadcast once per batch
    bundle0 = None
    if rank == 0:
        gb = torch.Generator().manual_seed(5)
        bundle0 = {"style": torch.randn(1, 192, generator=gb).to(dev),
                   "emo_vec": (torch.randn(1, D_model, generator=torch.Generator().manual_seed(6)) * 0.1).to(dev),
                   # what the prompt-side stages hand over for the flow-matching decoder (indextts/infer_v2_5.py:641-667)
                   "ref_mel": (torch.randn(1, n_mels, args.prompt_frames, generator=gb) * 2 - 4).to(dev),
                   "prompt_condition": torch.randn(1, args.prompt_frames, 512, generator=gb).to(dev)}

    def one_step(record):
        bundle = D.broadcast_speaker_bundle(bundle0, src=0, device=dev) if dist is not None else bundle0
        wav16 = eng.step(text, langs, mel, bundle, n_gen, record)
        return D.gather_waveform_tensor(wav16, mine, n_total, dst=0) if dist is not None else wav16

    def barrier():
        if not stub:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    for i in range(args.warmup):
        tw = time.perf_counter()
        one_step(False)
        if not stub:
            torch.cuda.synchronize()
        if rank == 0:
            log(f"[bench] warmup step {i}: {time.perf_counter() - tw:.2f}s")
    def steps_overlapped(n_steps):
        """The same n_steps steps as a two-stage software pipeline.  Stage 1 (decode of step k + 1) runs on `s_dec` from a worker thread
        -- the device decode loop blocks its host thread, and HIP's current device / torch's current stream are per thread -- stage 2
        (render of step k) on the main stream; the hand-over is an event the main stream waits on.  Nothing is skipped: n_steps decodes
        and n_steps renders complete before the closing barrier; with n_steps = 1 this degenerates to the sequential step."""
        import threading
        s_dec = None if stub else torch.cuda.Stream(device=dev, priority=-1)
        slot = {}

        def decode_worker(k, bundle):
            try:
                if stub:
                    slot[k] = (eng.decode(text, langs, bundle, n_gen, True), None)
                    return
                torch.cuda.set_device(dev)
                tw = time.perf_counter()
                with torch.cuda.stream(s_dec):
                    codes = eng.decode(text, langs, bundle, n_gen, True)
                    ev = torch.cuda.Event()
                    ev.record(s_dec)
                slot[k] = (codes, ev)
                if rank == 0:
                    log(f"[bench] overlap: decode of step {k} took {time.perf_counter() - tw:.2f}s of host wall time")
            except BaseException as e:      # surfaced by the main thread
                slot[k] = e

        def start(k):
            # the speaker bundle of step k is broadcast by the main thread (the only thread that issues collectives), then handed over
            bundle = D.broadcast_speaker_bundle(bundle0, src=0, device=dev) if dist is not None else bundle0
            if not stub:
                torch.cuda.current_stream().synchronize()
            th = threading.Thread(target=decode_worker, args=(k, bundle), name=f"decode-{k}")
            th.start()
            return th, bundle

        out = None
        th, bundle = start(0)
        for k in range(n_steps):
            th.join()
            got = slot.pop(k)
            if isinstance(got, BaseException):
                raise got
            codes, ev = got
            cur_bundle = bundle
            if k + 1 < n_steps:
                th, bundle = start(k + 1)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
                codes.record_stream(torch.cuda.current_stream())
            tw = time.perf_counter()
            wav16 = eng.render(codes, mel, cur_bundle, n_gen, True)
            out = D.gather_waveform_tensor(wav16, mine, n_total, dst=0) if dist is not None else wav16
            if rank == 0 and not stub:
                log(f"[bench] overlap: render of step {k} enqueued / finished on the host after {time.perf_counter() - tw:.2f}s")
        return out

    barrier()
    t0 = time.perf_counter()
    if args.overlap and args.steps > 1:
        wavs = steps_overlapped(args.steps)
    else:
        for _ in range(args.steps):
            wavs = one_step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        assert wavs.shape == (n_total, t_mel * HOP), wavs.shape       # every utterance of the batch arrived on rank 0

    audio_per_step = n_total * (t_mel * HOP) / SR
    value = audio_per_step * args.steps / elapsed
    if rank == 0:
        out = {
            "metric": METRIC if not stub else "STUB ENGINE (launcher test) -- not a measurement",
            "value": value, "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None,
            "dtype": ("bf16 (GPT and s2mel GEMM operands / K,V / attention probabilities; f32 accumulate, residual streams, norms) + "
                      "f32 (codec decode, length regulator, BigVGAN)") if args.precision == "bf16" else "f32",
            "data": "synthetic (seeded random-init weights of the IndexTTS-2.5 architecture; synthetic text ids, conditioning "
                    "vectors, prompt mel / prompt condition" + ("" if not args.no_s2mel else " and mel") +
                    "; EOS suppressed so every row decodes all tokens)",
            "rtf": 1.0 / value * n_total,          # wall seconds per audio second of ONE utterance stream
            "engine": eng.name,
            "config": {"workload": (f"IndexTTS-2.5 codes-to-waveform path, {n_total} utterances x {n_text} text tokens -> GPT decode of "
                                    f"{n_gen} speech tokens (top-k 30, top-p 0.8, T 0.8, rep-penalty 10, num_beams 1) -> "
                                    + (f"semantic-codec decode + length regulator + 25-step CFG flow matching (DiT 13 x 512, "
                                       f"{args.prompt_frames}-frame speaker prompt) -> " if not args.no_s2mel else
                                       "[s2mel skipped: synthetic mel] -> ")
                                    + f"BigVGAN-v2 22 kHz on {t_mel}-frame mels (BASELINE.json configs[2]), {B} utterances on each of "
                                      f"{world} GPU(s); prompt encoders / text front end not included"),
                       "global_batch": n_total, "per_gpu_batch": B, "text_tokens": n_text, "gen_tokens": n_gen,
                       "mel_frames": t_mel, "audio_seconds_per_utt": t_mel * HOP / SR, "parallelism": f"utterance-dp{world}",
                       "use_hipgraph": not args.no_graph,
                       "step_overlap": bool(args.overlap and args.steps > 1)},
        }
        if stub:       # launcher test: every utterance of the batch reached rank 0, in utterance order
            out["stub_rows_ok"] = bool(torch.equal(wavs[:, 0].to(torch.int64), text_all[:, 0].to(torch.int64) % 97))
        if not stub:
            out.update(gpu_report(args, eng, B, n_text, n_gen, t_mel))
            log("[bench] GPU result:", json.dumps({k: out[k] for k in ("value", "ms_per_step", "roofline", "stages")}))
            if not args.no_extras and world == 1:
                t_x = time.perf_counter()
                try:
                    out["stages"].update(eng.extra_modes(text, langs, bundle0["style"], bundle0["emo_vec"], t_mel=t_mel))
                    h3 = out["stages"].get("bigvgan_f16x3_mode")
                    if h3:       # what the headline would be with this mode promoted (NOT the reported value)
                        ms_alt = out["ms_per_step"] - out["stages"]["bigvgan_ms_per_step"] + h3["ms_per_step"]
                        h3["audio_seconds_per_sec_if_promoted"] = audio_per_step / (ms_alt * 1e-3)
                except Exception as e:
                    out["stages"]["extras_error"] = repr(e)
                log(f"[bench] extra modes (beam-3, f32) took {time.perf_counter() - t_x:.1f}s")
            if not args.no_cpu_baseline and world == 1:
                t_cpu = time.perf_counter()
                try:
                    out["cpu_baseline"] = cpu_baseline(eng.gsd, eng.gcfg, eng.bsd, eng.bh, n_text, n_gen, t_mel,
                                                       args.cpu_threads or usable_cores(),
                                                       None if args.no_s2mel else (args.prompt_frames + t_mel, args.prompt_frames))
                except Exception as e:      # never lose the GPU line because the baseline leg failed
                    out["cpu_baseline"] = {"error": repr(e)}
                log(f"[bench] cpu_baseline leg took {time.perf_counter() - t_cpu:.1f}s")
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def gpu_report(args, eng, B, n_text, n_gen, t_mel):
    """roofline (dominant kernel) + stage split from rank 0's HIP-event records of the timed steps."""
    prof_acc, gpt_t, gcfg = eng.prof_acc, eng.gpt_t, eng.gcfg
    conv = prof_acc.get("conv1d_mfma", dict(ms=1e-9, launches=1, flops=0.0, bytes=0.0))
    achieved = conv["flops"] / (conv["ms"] * 1e-3) / 1e12
    # HBM traffic of the same kernel: PMC counters cannot be read from inside this process, so the figure comes from
    # the committed rocprofv3 --pmc summary of the same forward (tools/pmc_bench_traffic.sh), if it matches this shape.
    traffic, traffic_src = None, None
    tp = os.path.join(ROOT, "profiles", "conv_traffic.json")
    if os.path.exists(tp):
        tj = json.load(open(tp))
        if tj.get("B") == B and tj.get("mel_frames") == t_mel:
            traffic = tj["hbm_bytes_per_conv_dispatch"]
            traffic_src = tj.get("source", "profiles/conv_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)")
    D, L, V = gcfg["model_dim"], gcfg["layers"], gcfg["number_mel_codes"]
    esz = 2 if args.precision == "bf16" else 4
    n_dec = max(1, gpt_t["steps"] - args.steps)                # decode steps (first token comes from prefill)
    ms_tok = gpt_t["decode_ms"] / n_dec
    ctx_avg = (n_text + 6) + n_gen / 2.0
    bytes_step = (12 * D * D * L + D * V) * esz + B * 2 * L * D * ctx_avg * esz
    s_pre = n_text + 6                                         # 3 cond + text + start/stop text + start mel
    prefill_flops = 2.0 * (12 * D * D * L) * B * s_pre + 4.0 * L * D * s_pre * s_pre * B / 2 + 2.0 * D * V * B
    prefill_tflops = prefill_flops / (gpt_t["prefill_ms"] / args.steps * 1e-3) / 1e12
    stages = {
        "gpt_prefill_ms_per_step": gpt_t["prefill_ms"] / args.steps,
        "gpt_prefill_tflops": prefill_tflops,          # whole prefill pass (GEMMs + attention + LayerNorms) per wall time
        "gpt_prefill_mfma_frac": prefill_tflops / (PEAK_BF16_MFMA_TFLOPS if args.precision == "bf16" else PEAK_F32_MFMA_TFLOPS),
        "gpt_decode_ms_per_step": gpt_t["decode_ms"] / args.steps,
        "gpt_decode_ms_per_token": ms_tok,
        "gpt_decode_algorithmic_GBps": bytes_step / (ms_tok * 1e-3) / 1e9,
        "gpt_decode_hbm_frac": bytes_step / (ms_tok * 1e-3) / 1e9 / PEAK_HBM_GBPS,
        "s2mel": None if eng.s2 is None else {
            "codec_regulator_ms_per_step": eng.s2_t["codec_regulator_ms"] / args.steps,
            "cfm_ms_per_step": eng.s2_t["cfm_ms"] / args.steps,        # 25 Euler steps x CFG batch-2 estimator, host prep included
            "cfm_estimator_ms_per_step": eng.s2_t["estimator_ms"] / args.steps,
            "cfm_gemm_ms_per_step": eng.s2_t["gemm_ms"] / args.steps,
            "cfm_gemm_tflops": eng.s2_t["gemm_flops"] / max(1e-9, eng.s2_t["gemm_ms"] * 1e-3) / 1e12,
            "cfm_gemm_mfma_frac": eng.s2_t["gemm_flops"] / max(1e-9, eng.s2_t["gemm_ms"] * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
            "cfm_attention_ms_per_step": eng.s2_t["attention_ms"] / args.steps,
            "cfm_attention_tflops": eng.s2_t["attention_flops"] / max(1e-9, eng.s2_t["attention_ms"] * 1e-3) / 1e12,
            "cfm_elementwise_ms_per_step": (eng.s2_t["estimator_ms"] - eng.s2_t["gemm_ms"] - eng.s2_t["attention_ms"]) / args.steps,
            "prompt_frames": args.prompt_frames, "euler_steps": 25, "cfg_rate": 0.7},
        "bigvgan_ms_per_step": sum(v["ms"] for v in prof_acc.values()) / args.steps,
        "bigvgan_kernels": {k: dict(ms_per_step=v["ms"] / args.steps, launches_per_step=v["launches"] // args.steps,
                                    tflops=(v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0),
                                    GBps=(v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0))
                            for k, v in prof_acc.items()},
    }
    roofline = {"bound": "mfma", "kernel": "conv_mfma_kernel (BigVGAN Conv1d implicit GEMM, v_mfma_f32_32x32x2_f32)",
                "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                "traffic_source": traffic_src, "algorithmic_bytes_per_launch": conv["bytes"] / max(1, conv["launches"]),
                "launches_per_step": conv["launches"] // max(1, args.steps),
                "avg_launch_ms": conv["ms"] / max(1, conv["launches"])}
    return {"roofline": roofline, "stages": stages}


if __name__ == "__main__":
    main()
