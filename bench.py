#!/usr/bin/env python3
"""Headline benchmark: audio-seconds generated per wall-second, IndexTTS-2.5 hot path, 64-utterance x 128-token batch.

One "step" = one pass of the hot path over the synthetic 64-utterance batch (BASELINE.json configs[2]):
    speaker-bundle broadcast (RCCL when N > 1)
    -> GPT speech-token decode of this rank's utterances (prefill + 560 sampled tokens each, top-k 30 / top-p 0.8 / T 0.8 /
       repetition penalty 10, bf16 weights + KV)
    -> codes -> mel: semantic-codec decode, length regulator, 25-step classifier-free-guidance flow matching over
       [speaker prompt | int(2 * n_tokens * 1.72) target frames] (indextts/infer_v2_5.py:830-846) in `--s2mel-precision`: fp32 by
       default -- the reference runs this stage with autocast off even in its bf16 mode (infer_v2_5.py:827-828) -- so `value` is the
       fp32-CFM line; the bf16 mode (bf16 GEMM operands / attention, f32 accumulation) is timed right after on the same inputs and
       printed beside it (`value_by_s2mel_precision`), with its measured waveform error cited from tests/test_gpu_fullsize.py
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
X3_PRODUCTS = [6]                 # plane products per f32 product of the fp32x3 mode in this run (--x3-products -> engine option x3_products)
PEAK_HBM_GBPS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA (no sparsity)
SR, HOP = 22050, 256
EULER_STEPS = 25                  # diffusion steps of cfm.inference in the pipeline (indextts/infer_v2_5.py:843)
METRIC = "audio-seconds/sec (RTF) IndexTTS-2.5, 64-utt batch @1/2/4/8 MI355X"



def x3_waves_opt(args, stub):
    """waves per block of the fp32x3 GEMM the run used (engine option x3_waves; None outside that mode / in the launcher-test stub)"""
    if stub or args.no_s2mel or args.s2mel_precision != "fp32x3":
        return None
    from indextts_amd import _lib
    return int(_lib.get_option("x3_waves"))

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
    The two stages whose checked form is slower on CPU than the reference's own modules (index-arithmetic resamplers, written-out attention) run
    in the oracles' TIMING_MODE (strided depthwise convolutions, fused CPU attention: the forms the reference calls), so that the port is not a
    slower baseline than the reference: measured beside the reference's classes in the build container, profiles/r03z_cpu/reference_cpu_timing.log.
    """
    from oracle import bigvgan_oracle as BO
    from oracle import gpt_oracle as GO
    from oracle import s2mel_oracle as SO
    BO.TIMING_MODE = SO.TIMING_MODE = True
    try:
        return _cpu_baseline_timed(BO, GO, SO, gpt_sd, gpt_cfg, bv_sd, bv_h, n_text, n_gen, t_mel, threads, s2mel_frames)
    finally:
        BO.TIMING_MODE = SO.TIMING_MODE = False


def _timed(f):
    t0 = time.perf_counter()
    f()
    return time.perf_counter() - t0


def _cpu_baseline_timed(BO, GO, SO, gpt_sd, gpt_cfg, bv_sd, bv_h, n_text, n_gen, t_mel, threads, s2mel_frames):
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
        t_frame = min(_timed(lambda: BO.bigvgan_forward(bv_sd, mel, bv_h)) for _ in range(2)) / frames        # first call creates the oneDNN primitives
        # s2mel: ONE Euler step (one CFG-stacked estimator call of the DiT 13 x 512 + WaveNet 8 x 512) at the bench's frame count,
        # x euler_steps; skipped (and left out of `value`) when the GPU line runs without the s2mel stage
        t_euler = None
        if s2mel_frames:
            scfg = SO.S2MelConfig()
            ssd = SO.synth_weights(scfg, 3)
            T, Tp = s2mel_frames
            z = torch.randn(1, scfg.in_channels, T, generator=g)
            pr, mu, st = (torch.randn(1, scfg.in_channels, Tp, generator=g), torch.randn(1, T, scfg.content_dim, generator=g),
                          torch.randn(1, scfg.style_dim, generator=g))
            t_euler = min(_timed(lambda: SO.cfm_solve_euler(ssd, scfg, z, torch.tensor([T]), pr, mu, st, 1, 0.7)) for _ in range(2))
    audio_s = t_mel * HOP / SR
    cpu_time = t_prefill + n_gen * t_tok + t_mel * t_frame + (EULER_STEPS * t_euler if t_euler is not None else 0.0)
    res = {"value": audio_s / cpu_time, "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
           "sample": f"oracle (torch fp32 CPU restatement of the reference path; resamplers and attention in the forms the reference calls on CPU): GPT 1 utt x {n_text} text tokens, prefill + "
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
        self.voc = bigvgan.BigVGAN(self.bh, device=dev, conv_mode=None if args.bigvgan_conv == "f32" else args.bigvgan_conv)
        self.voc.load_state_dict(self.bsd)
        self.voc.to(dev)
        self.prof_acc = {}
        self.n_prof = 0                    # timed steps that carried per-launch HIP-event profiling (the stage split's denominator)
        self.gpt_t = {"prefill_ms": 0.0, "decode_ms": 0.0, "steps": 0}
        self.s2 = None
        self.s2_t = self.new_stage_acc()
        if not args.no_s2mel:
            from indextts_amd import codec
            self.codec = codec.EnhancedCodec(**synth.CODEC_V2, device=dev)
            self.codec.load_state_dict(synth.codec_weights(seed=1234))
            self.s2_args = dict(synth.S2MEL_V2, length_regulator=synth.REGULATOR_V2)
            self.s2 = self.build_s2(args.s2mel_precision)
            self._ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        self.gen_kw = dict(do_sample=True, top_p=0.8, top_k=30, temperature=0.8, num_beams=1, repetition_penalty=10.0,
                           length_penalty=0.0)
        if rank == 0:
            log(f"[bench] weights synthesised + packed + uploaded in {time.perf_counter() - t_load:.1f}s")

    def build_s2(self, precision):
        from indextts_amd import s2mel, synth
        s2 = s2mel.MyModel(self.s2_args, precision=precision, device=self.dev)
        s2.models["cfm"].load_state_dict(synth.s2mel_weights(seed=1234))
        s2.models["length_regulator"].load_state_dict(synth.regulator_weights(seed=1234))
        return s2

    def set_profiling(self, on):
        """Per-launch HIP events around every vocoder launch and every CFM GEMM / attention launch (a few thousand pairs per step):
        switched on for the first `--profile-steps` timed steps only, so the headline does not pay for its own instrumentation."""
        self.voc.set_profiling(bool(on))
        if self.s2 is not None:
            self.s2.models["cfm"].set_profiling(bool(on))

    def new_stage_acc(self):
        return {"codec_regulator_ms": 0.0, "cfm_ms": 0.0, "gemm_ms": 0.0, "gemm_flops": 0.0, "attention_ms": 0.0,
                "attention_flops": 0.0, "estimator_ms": 0.0, "launches": 0, "attention_launches": 0}

    def step(self, text, langs, mel, bundle, n_gen, record):
        """-> int16 waveforms (b, T*256) of this rank's utterances"""
        # record: True = a timed step with per-launch profiling, "gpt" = a timed step without it (only the GPT stage's own timers,
        # which cost nothing, are accumulated), False = warmup
        if record is True:
            self.n_prof += 1
        return self.render(self.decode(text, langs, bundle, n_gen, bool(record)), mel, bundle, n_gen, record is True)

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

    def render(self, codes, mel, bundle, n_gen, record, s2=None, acc=None):
        """codes -> codec decode -> length regulator -> 25-step CFG flow matching -> BigVGAN -> int16, on torch's current stream.
        s2 / acc: another s2mel engine (the second precision) and the stage accumulator its profile goes to"""
        B = codes.shape[0]
        style = bundle["style"]
        s2 = s2 if s2 is not None else self.s2
        if s2 is not None:
            # codes -> content features -> 25-step CFG flow matching -> mel (indextts/infer_v2_5.py:830-846), all on the engine
            cfm = s2.models["cfm"]
            self._ev[0].record()
            S_infer = self.codec.decode(codes, code_lens=[n_gen] * B)
            target = [int(2 * n_gen * 1.72)] * B
            cond = s2.models["length_regulator"](S_infer, ylens=torch.tensor(target), n_quantizers=3, f0=None,
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
                t = acc if acc is not None else self.s2_t
                t["codec_regulator_ms"] += self._ev[0].elapsed_time(self._ev[1])
                t["cfm_ms"] += self._ev[1].elapsed_time(self._ev[2])
                t["gemm_ms"] += pr["gemm"]["ms"]
                t["gemm_flops"] += pr["gemm"]["flops"]
                t["attention_ms"] += pr["attention"]["ms"]
                t["attention_flops"] += 4.0 * cfm.hidden_dim * (2 * B) * float(total[0]) ** 2 * pr["attention"]["launches"]
                t["estimator_ms"] += pr["estimator_calls"]["ms"]
                t["launches"] += pr["gemm"]["launches"]            # GEMM launches only: `roofline.avg_launch_ms` is per launch of THAT kernel
                t["attention_launches"] += pr["attention"]["launches"]
        chunk = self.args.bigvgan_chunk or B
        outs = []
        for b0 in range(0, B, chunk):
            outs.append(self.voc(mel[b0:b0 + chunk]))
            if record and acc is None:
                for k, v in self.voc.profile().items():
                    a = self.prof_acc.setdefault(k, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
                    for kk in a:
                        a[kk] += v[kk]
        wav = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
        return torch.clamp(32767.0 * wav[:, 0], -32767.0, 32767.0).to(torch.int16)      # infer_v2_5.py:855, :897-898

    # short untimed extras (rank 0, N = 1): the cost of the other GPT modes at the bench shape
    def extra_modes(self, text, langs, style, emo_vec, n_tok=32, t_mel=None, n_gen=None):
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
            del v3, w3
            # ... and the vocoder's other exact-operand conv mode (f32 MFMA <-> bf16 x 3), one forward at the bench shape
            other = "f32" if self.args.bigvgan_conv != "f32" else "bf16x3"
            vo = bigvgan.BigVGAN(self.bh, device=self.dev, conv_mode=None if other == "f32" else other)
            vo.load_state_dict(self.bsd)
            vo.to(self.dev)
            vo.set_profiling(True)
            wo = vo(mel)
            wo = vo(mel)
            po = vo.profile()
            out["bigvgan_other_conv_mode"] = {"conv": other, "ms_per_step": sum(v["ms"] for v in po.values()), "conv_ms_per_step": po["conv1d_mfma"]["ms"],
                                              "conv_tflops_f32_equivalent": po["conv1d_mfma"]["flops"] / (po["conv1d_mfma"]["ms"] * 1e-3) / 1e12,
                                              "rms_vs_headline_mode": float((wo.float() - ref.float()).pow(2).mean().sqrt()),
                                              "signal_rms": float(ref.float().pow(2).mean().sqrt())}
            del vo, wo, ref
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
        if n_gen:
            # the reference's DEFAULT generation mode (3-beam beam-sample) over the headline's full token count: what the decode stage of the
            # timed step would cost with it (the headline decodes with num_beams = 1, the mode BASELINE.json's configs name: "top-p sampling")
            try:
                self.model.inference_speech(None, text, langs=langs, emo_vec=emo_vec, campplus_embedding=style, max_generate_length=n_gen, **kw)
                t = self.model.last_timing
                out["gpt_beam3_full_length"] = {"prefill_ms": t["prefill_ms"], "decode_ms": t["decode_ms"], "steps": t["steps"],
                                                "ms_per_token": t["decode_ms"] / max(1, t["steps"] - 1)}
            except Exception as e:
                out["gpt_beam3_full_length"] = {"error": repr(e)}
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
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"], help="GPT precision (bf16 = the reference's own GPU mode)")
    ap.add_argument("--s2mel-precision", default="fp32x3", choices=["fp32", "fp32x3", "bf16"],
                    help="precision of the flow-matching stage in the timed steps.  The reference computes the stage in fp32 (autocast off "
                         "around s2mel, infer_v2_5.py:827-828).  fp32x3 (default, carries `value`; ruled admissible in VERDICT r3): f32 "
                         "activations and weights, every GEMM AND attention operand carried EXACTLY as three bf16 planes (24 significand bits), "
                         "--x3-products plane products per f32 product on the bf16 matrix pipe, f32 accumulation -- error against an f64 result not "
                         "above the native f32-MFMA kernels' (tests/test_gpu_gemm_x3.py, tests/test_gpu_attn_x3.py); softmax, norms, every "
                         "element-wise stage: the f32 code.  fp32: the "
                         "native f32-MFMA kernels; bf16: 16-bit operands (not admissible as the headline).  The other modes are timed after the "
                         "headline on the same inputs and printed beside it (`value_by_s2mel_precision`)")
    ap.add_argument("--x3-products", type=int, default=6, choices=[6, 8],
                    help="plane products per f32 product of the fp32x3 mode (engine option x3_products): 6 = hh, hm, mh, hl, lh, mm (the two dropped "
                         "terms are 2^-24 |ab| each; VERDICT r3's condition -- GPU error vs f64 <= the native f32 kernel's on every s2mel GEMM shape, "
                         "solve-level tests unchanged -- is met: tests/test_gpu_gemm_x3.py, profiles/r04b), 8 = every term down to 2^-24 |ab|")
    ap.add_argument("--profile-steps", type=int, default=1, help="timed steps that carry per-launch HIP-event profiling (stage split, "
                    "roofline); the remaining timed steps run without the instrumentation")
    ap.add_argument("--alt-steps", type=int, default=2, help="timed steps of each other s2mel precision (after the headline)")
    ap.add_argument("--no-configs", action="store_true", help="skip the timed lines of BASELINE.json's other configs")
    ap.add_argument("--no-shards", action="store_true", help="skip the timed per-rank shards of the 2 / 4 / 8-GPU runs and the 15 s-prompt line")
    ap.add_argument("--bigvgan-chunk", type=int, default=0, help="utterances per BigVGAN launch group (0 = all)")
    ap.add_argument("--bigvgan-conv", default="bf16x3", choices=["bf16x3", "f32"],
                    help="BigVGAN resblock convs: bf16x3 = every f32 operand exactly as three bf16 planes, six plane products on the bf16 matrix pipe, "
                         "f32 accumulation (the fp32x3 arithmetic of the flow-matching stage; error vs f64 not above the f32-MFMA kernel's, "
                         "tests/test_gpu_bigvgan_x3.py); f32 = the f32 MFMA kernel everywhere")
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

    X3_PRODUCTS[0] = args.x3_products
    if not stub:
        from indextts_amd import _lib
        _lib.set_option("x3_products", args.x3_products)
    eng = (StubEngine if stub else HipEngine)(args, dev, rank)
    n_text, n_gen = args.text_tokens, args.gen_tokens
    t_mel = int(2 * n_gen * 1.72)
    n_total = args.utts * (world if args.weak else 1)
    # the whole synthetic batch is generated identically on every rank (seeded), then sharded: strong scaling hands each
    # rank its LPT share of the SAME 64 utterances (all 128 text tokens long here, so the shares are equal)
    g = torch.Generator().manual_seed(100)
    n_text_ids, n_mels, D_model = (12000, 80, 1280) if stub else (eng.gcfg["number_text_tokens"], eng.bh["num_mels"], eng.gcfg["model_dim"])
    text_all = torch.randint(2, n_text_ids, (n_total, n_text), generator=g)
    shards = [D.shard_utterances(n_total, r, world, lengths=[n_text] * n_total) for r in range(world)]     # the same table on every rank
    mine = shards[rank]
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
        return D.gather_waveform_tensor(wav16, mine, n_total, dst=0, shards=shards) if dist is not None else wav16

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
            out = D.gather_waveform_tensor(wav16, mine, n_total, dst=0, shards=shards) if dist is not None else wav16
            if rank == 0 and not stub:
                log(f"[bench] overlap: render of step {k} enqueued / finished on the host after {time.perf_counter() - tw:.2f}s")
        return out

    step_ev = [] if stub else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    n_prof = 0 if stub else max(0, min(args.profile_steps, args.steps))
    barrier()
    t0 = time.perf_counter()
    if args.overlap and args.steps > 1:
        if not stub:
            eng.set_profiling(True)
            eng.n_prof = args.steps
        wavs = steps_overlapped(args.steps)
    else:
        for k in range(args.steps):
            if not stub:
                if k == 0 or k == n_prof:
                    eng.set_profiling(k < n_prof)
                step_ev[k][0].record()
            wavs = one_step(True if k < n_prof else "gpt")
            if not stub:
                step_ev[k][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = [] if (stub or (args.overlap and args.steps > 1)) else [a.elapsed_time(b) for a, b in step_ev]
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
            "dtype": ((lambda d: d if args.bigvgan_conv == "f32" else d.replace("BigVGAN)", "BigVGAN outside the resblock convs) + bf16x3 (BigVGAN resblock "
                       "convs of the >= 96-channel stages: f32 operands exactly as three bf16 planes, six plane products, f32 accumulation)")
                       .replace("BigVGAN:", "BigVGAN outside the resblock convs:"))(("bf16 (GPT GEMM operands / KV cache; f32 accumulate, residual stream, norms: the reference's own GPU mode) + "
                       if args.precision == "bf16" else "f32 (GPT) + ")
                      + ("f32 (codec decode, length regulator, flow matching, BigVGAN: the reference runs these with autocast off)"
                         if args.s2mel_precision == "fp32" or args.no_s2mel else
                         f"fp32x3 (flow matching: f32 activations / weights / accumulation, GEMM and attention products on the bf16 matrix pipe with "
                         f"every f32 operand carried exactly as three bf16 planes = 24 significand bits, {args.x3_products} plane products per f32 "
                         f"product; softmax, norms, element-wise stages f32) + f32 (codec decode, length regulator, BigVGAN)"
                         if args.s2mel_precision == "fp32x3" else
                         "bf16 (s2mel GEMM operands / K,V / attention probabilities; f32 accumulate, residual streams, norms) + "
                         "f32 (codec decode, length regulator, BigVGAN)"))),
            "data": "synthetic (seeded random-init weights of the IndexTTS-2.5 architecture; synthetic text ids, conditioning "
                    "vectors, prompt mel / prompt condition" + ("" if not args.no_s2mel else " and mel") +
                    "; EOS suppressed so every row decodes all tokens)",
            "rtf": 1.0 / value * n_total,          # wall seconds per audio second of ONE utterance stream
            "engine": eng.name,
            "config": {"workload": (f"IndexTTS-2.5 codes-to-waveform path, {n_total} utterances x {n_text} text tokens -> GPT decode of "
                                    f"{n_gen} speech tokens (top-k 30, top-p 0.8, T 0.8, rep-penalty 10, num_beams 1) -> "
                                    + (f"semantic-codec decode + length regulator + 25-step CFG flow matching (DiT 13 x 512, "
                                       f"{args.prompt_frames}-frame speaker prompt, {args.s2mel_precision}) -> " if not args.no_s2mel else
                                       "[s2mel skipped: synthetic mel] -> ")
                                    + f"BigVGAN-v2 22 kHz on {t_mel}-frame mels (BASELINE.json configs[2]), {B} utterances on each of "
                                      f"{world} GPU(s); prompt encoders / text front end not included"),
                       "global_batch": n_total, "per_gpu_batch": B, "text_tokens": n_text, "gen_tokens": n_gen,
                       "mel_frames": t_mel, "audio_seconds_per_utt": t_mel * HOP / SR, "parallelism": f"utterance-dp{world}",
                       "gpt_precision": args.precision, "s2mel_precision": None if args.no_s2mel else args.s2mel_precision,
                       "x3_products": args.x3_products if (not args.no_s2mel and args.s2mel_precision == "fp32x3") else None,
                       "x3_gemm_waves_per_block": x3_waves_opt(args, stub),
                       "use_hipgraph": not args.no_graph,
                       "step_overlap": bool(args.overlap and args.steps > 1)},
        }
        if stub:       # launcher test: every utterance of the batch reached rank 0, in utterance order
            out["stub_rows_ok"] = bool(torch.equal(wavs[:, 0].to(torch.int64), text_all[:, 0].to(torch.int64) % 97))
        if not stub:
            out.update(gpu_report(args, eng, B, n_text, n_gen, t_mel))
            out["step_ms"] = step_ms            # per timed step (HIP events): the first --profile-steps carry the per-launch instrumentation
            st = out["stages"]
            if step_ms and len(step_ms) > eng.n_prof > 0:
                st["profiled_step_ms"] = sum(step_ms[:eng.n_prof]) / eng.n_prof
                st["unprofiled_step_ms"] = sum(step_ms[eng.n_prof:]) / (len(step_ms) - eng.n_prof)
            # the two north_star hot paths alone (GPT + BigVGAN, both at the reference's precision), from the stage timers
            hot_ms = st["gpt_prefill_ms_per_step"] + st["gpt_decode_ms_per_step"] + st["bigvgan_ms_per_step"]
            out["value_north_star_paths"] = {"value": audio_per_step / (hot_ms * 1e-3), "ms_per_step": hot_ms,
                                             "what": "GPT decode + BigVGAN only (the two hot paths north_star names), from the stage timers"}
            if not args.no_s2mel:
                out["value_by_s2mel_precision"] = {args.s2mel_precision: {"value": value, "ms_per_step": out["ms_per_step"], "steps": args.steps}}
            log("[bench] GPU result:", json.dumps({k: out[k] for k in ("value", "ms_per_step", "roofline", "stages")}))
            if not args.no_s2mel and not args.no_extras and world == 1 and args.alt_steps > 0 and not (args.overlap and args.steps > 1):
                # the other s2mel precision on the same inputs: --alt-steps timed steps (one of them profiled), outside the headline's timed region
                t_x = time.perf_counter()
                try:
                    for alt in [p for p in ("fp32", "fp32x3", "bf16") if p != args.s2mel_precision]:
                        out["value_by_s2mel_precision"][alt] = alt_precision_leg(args, eng, alt, text, langs, mel, bundle0, n_gen, audio_per_step, st)
                except Exception as e:
                    out["value_by_s2mel_precision"]["error"] = repr(e)
                log(f"[bench] second s2mel precision took {time.perf_counter() - t_x:.1f}s")
                out["s2mel_precision_note"] = (
                    "`value` is the line with the flow-matching stage in " + args.s2mel_precision + "; the reference computes it in fp32 "
                    "(indextts/infer_v2_5.py:827-828).  The bf16 mode's error at this shape after the full 25 steps (mel and BigVGAN "
                    "waveform RMS vs the f32 mode) is measured by tests/test_gpu_fullsize.py::test_cfm_bf16_error_after_25_steps_mel_and_waveform "
                    "and recorded in DESIGN.md section 9; it is above north_star's 1e-4 waveform bar, so the bf16 line is not the headline")
            if not args.no_extras and world == 1:
                t_x = time.perf_counter()
                try:
                    out["stages"].update(eng.extra_modes(text, langs, bundle0["style"], bundle0["emo_vec"], t_mel=t_mel, n_gen=n_gen))
                    b3 = out["stages"].get("gpt_beam3_full_length") or {}
                    if "decode_ms" in b3:    # the headline step with its decode stage replaced by the measured 3-beam one (derived, NOT the reported value)
                        ms_b3 = (out["ms_per_step"] - out["stages"]["gpt_prefill_ms_per_step"] - out["stages"]["gpt_decode_ms_per_step"]
                                 + b3["prefill_ms"] + b3["decode_ms"])
                        out["value_if_num_beams_3"] = {"value": audio_per_step / (ms_b3 * 1e-3), "ms_per_step": ms_b3,
                                                       "what": "the timed step with its GPT stage replaced by the measured 3-beam beam-sample decode of the "
                                                               "same batch and length (the reference's default generation mode); derived from two measurements"}
                    h3 = out["stages"].get("bigvgan_f16x3_mode")
                    if h3:       # what the headline would be with this mode promoted (NOT the reported value)
                        ms_alt = out["ms_per_step"] - out["stages"]["bigvgan_ms_per_step"] + h3["ms_per_step"]
                        h3["audio_seconds_per_sec_if_promoted"] = audio_per_step / (ms_alt * 1e-3)
                except Exception as e:
                    out["stages"]["extras_error"] = repr(e)
                log(f"[bench] extra modes (beam-3, f32) took {time.perf_counter() - t_x:.1f}s")
            if not args.no_extras and not args.no_configs and not args.no_s2mel and world == 1:
                t_x = time.perf_counter()
                try:
                    eng.set_profiling(False)
                    out["stages"]["configs"] = configs_leg(args, eng, bundle0, dev)
                except Exception as e:
                    out["stages"]["configs"] = {"error": repr(e)}
                log(f"[bench] other BASELINE configs took {time.perf_counter() - t_x:.1f}s")
            if not args.no_extras and not args.no_shards and not args.no_s2mel and world == 1 and not args.weak:
                t_x = time.perf_counter()
                try:
                    eng.set_profiling(False)
                    sh = shard_leg(args, eng, text, langs, mel, bundle0, n_gen, t_mel)
                    out["stages"].setdefault("configs", {}).update(sh["configs"])
                    out.update(sh["projections"])
                except Exception as e:
                    out["stages"].setdefault("configs", {})["shards_error"] = repr(e)
                log(f"[bench] per-rank shards of the 2 / 4 / 8-GPU runs + 15 s prompt took {time.perf_counter() - t_x:.1f}s")
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


def configs_leg(args, eng, bundle0, dev):
    """Timed lines of BASELINE.json's OTHER configs and of the one published comparison point (single-utterance RTF), through the
    PRODUCT pipeline classes' own `_synthesize` (GPT batch -> stop-token trim -> codes -> mel -> ragged BigVGAN batch -> float waveforms
    on the host), with this process's engines: untimed extras beside the headline (`stages.configs`), one warm call + one timed call
    each; EOS is suppressed, so every row decodes `gen_tokens`.  s2mel runs in the headline's precision (--s2mel-precision).
      configs[0]  IndexTTS-1.5, one utterance x 32 text tokens, greedy, 96 codes -> latent pass -> v1 vocoder (`config0_leg`)
      configs[1]  IndexTTS-2.5, 8 utterances x 64 text tokens, reference-default 3-beam beam-sample (top-p 0.8 / top-k 30 / T 0.8), 350 codes
      configs[3]  IndexTTS-2, 16 utterances x 64 text tokens at 24 x 1280: emotion path (merge_emovec: two Conformer + Perceiver passes
                  over 15 s prompts), speaker latents, 34 conditioning tokens, 700 codes (50 / s), teacher-forced latent pass,
                  gpt_layer + vq2emb -> regulator -> CFM -> BigVGAN
      configs[4]  IndexTTS-2.5 long-form: 2 000 characters = 17 segments of <= 120 text tokens decoded as ONE batch, duration_factor 1.5,
                  plus the vocoder alone as an exact overlap-save stream (256-frame chunks, receptive-field halo) over the same mels
      b1_rtf      one utterance, 40 text tokens (~80 characters) -> 200 codes (8 s): beside the reference README's RTF table (other hardware)
      ragged_b64  the headline's 64 x 128-token batch with utterance lengths spread over 280..560 codes (per-row caps: every request of a
                  merged batch keeps its own max_mel_tokens): audio-seconds counted on the ACTUAL lengths; the decode stage alone with row
                  compaction off / on (finished rows leave the running batch in 8-row buckets; backends/trt/pipeline/pipeline.py:459-548 is
                  the reference design for continuous batching)
    """
    import warnings
    from indextts_amd import gpt, infer_v2, infer_v2_5, s2mel, synth
    out = {}

    class NoFrontend:                      # `_synthesize` takes token tensors and a speaker bundle: no prompt / text front end involved
        pass

    def segments(n_seg, n_text, seed):
        g = torch.Generator().manual_seed(seed)
        return [torch.cat([torch.randint(2, 12000, (n_text,), generator=g).to(torch.int32), torch.ones(1, dtype=torch.int32)])
                for _ in range(n_seg)]

    def timed(tts, segs, bundle, emovec, df, gen):
        res = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")                  # "generation stopped due to exceeding max_mel_tokens": EOS is suppressed
                wavs = tts._synthesize(segs, [3] * len(segs), bundle, emovec, df, dict(gen), 120)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            audio = sum(int(w.shape[-1]) for w in wavs) / SR
            res = {"ms": dt * 1e3, "audio_seconds": audio, "audio_seconds_per_sec": audio / dt, "rtf_per_stream": dt / (audio / len(wavs)),
                   "stage_seconds": {k: round(float(v), 4) for k, v in tts.last_timing.items()}}
            gt = getattr(eng.model, "last_timing", None) or {}
            if "prefill_ms" in gt:                               # device events around the prefill launch train, which ends with the first sampled code
                res["time_to_first_token_ms"] = round(float(gt["prefill_ms"]), 3)
                res["decode_ms_per_token"] = round(float(gt["decode_ms"]) / max(1, int(gt.get("steps", 1))), 4)
        return res

    style, emo_vec = bundle0["style"], bundle0["emo_vec"]
    bundle = dict(bundle0, spk_cond_emb=torch.zeros(1, 4, 1024, device=dev))
    tts = infer_v2_5.IndexTTS2(cfg={"gpt": {"stop_mel_token": 8193}, "version": 2.5}, device=str(dev), frontend=NoFrontend(), gpt=eng.model,
                               bigvgan=eng.voc, semantic_codec=eng.codec, s2mel=eng.s2, codes_to_mel="engine")
    sample = dict(top_p=0.8, top_k=30, temperature=0.8, repetition_penalty=10.0)
    try:
        r = timed(tts, segments(8, 64, 301), bundle, emo_vec, 1.0, dict(sample, num_beams=3, max_mel_tokens=350))
        out["config1_v25_b8_beam3"] = dict(r, batch=8, text_tokens=64, gen_tokens=350, num_beams=3, mel_frames=int(2 * 350 * 1.72))
    except Exception as e:
        out["config1_v25_b8_beam3"] = {"error": repr(e)}
    try:
        r = timed(tts, segments(1, 40, 302), bundle, emo_vec, 1.0, dict(sample, num_beams=1, max_mel_tokens=200))
        r3 = timed(tts, segments(1, 40, 302), bundle, emo_vec, 1.0, dict(sample, num_beams=3, max_mel_tokens=200))
        out["b1_rtf"] = dict(r, batch=1, text_tokens=40, gen_tokens=200, num_beams=1, rtf_beam3=r3["rtf_per_stream"],
                             published_context={"reference README.md:474-483 (RTX 4090, IndexTTS-2, fp16)": 0.2065,
                                                "backends/trt/README.md:63-76 (TensorRT, RTX 4090)": 0.1365,
                                                "note": "other hardware, real checkpoints, prompt encoders included there: context, not a baseline"})
    except Exception as e:
        out["b1_rtf"] = {"error": repr(e)}
    try:
        segs = segments(17, 118, 303)                              # 2 000 chars -> ceil(tokens / 120) ~ 17 segments
        r = timed(tts, segs, bundle, emo_vec, 1.5, dict(sample, num_beams=1, max_mel_tokens=480))
        t_mel = int(2 * 480 * 1.72 * 1.5)
        mel = (torch.randn(1, 80, t_mel, generator=torch.Generator().manual_seed(9)) * 2 - 4).to(dev)
        eng.voc.forward_chunked(mel, chunk_frames=256)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(17):
            eng.voc.forward_chunked(mel, chunk_frames=256)
        torch.cuda.synchronize()
        out["config4_v25_longform"] = dict(r, segments=17, text_tokens=118, gen_tokens=480, duration_factor=1.5, mel_frames=t_mel,
                                           vocoder_streamed_ms=(time.perf_counter() - t0) * 1e3,
                                           vocoder_streamed_what="17 segments, each vocoded alone as an exact overlap-save stream of 256-frame chunks")
    except Exception as e:
        out["config4_v25_longform"] = {"error": repr(e)}
    try:
        g = torch.Generator().manual_seed(306)
        caps = torch.randint(280, 561, (64,), generator=g).tolist()
        segs = segments(64, 128, 307)
        text = torch.stack([t for t in segs]).to(dev)
        langs = torch.full((64,), 3, dtype=torch.long, device=dev)
        dec = {}
        for name, on in (("compaction_off", False), ("compaction_on", True)):
            eng.model.set_compaction(on, 8)
            for _ in range(2):
                eng.model.inference_speech(None, text, langs=langs, emo_vec=emo_vec, campplus_embedding=style, max_generate_length=560,
                                           do_sample=True, num_beams=1, row_max_new=caps, **sample)
            t = eng.model.last_timing
            dec[name] = {"decode_ms": t["decode_ms"], "steps": t["steps"], "row_steps": t["row_steps"], "compactions": t["compactions"]}
        eng.model.set_compaction(True, 8)
        r = timed(tts, segs, bundle, emo_vec, 1.0, dict(sample, num_beams=1, max_mel_tokens=560, row_max_new=caps))
        out["ragged_b64"] = dict(r, batch=64, text_tokens=128, code_lengths="uniform 280..560 (seeded), mean %.0f" % (sum(caps) / 64.0),
                                 decode_only=dec, decode_speedup=dec["compaction_off"]["decode_ms"] / max(1e-9, dec["compaction_on"]["decode_ms"]))
    except Exception as e:
        out["ragged_b64"] = {"error": repr(e)}
    try:
        # in-flight batching of the GPT stage: 256 utterances of ragged length on 64 decode slots, freed slots refilled from the waiting utterances
        # (UnifiedVoice.inference_speech_inflight: one session, every slot with its own cache position / step / budget), against the same
        # utterances as four drained batches of 64 (row compaction on in both)
        g = torch.Generator().manual_seed(308)
        n_utt = 256
        caps = torch.randint(120, 561, (n_utt,), generator=g).tolist()
        text = torch.stack(segments(n_utt, 128, 309)).to(dev)
        langs = torch.full((n_utt,), 3, dtype=torch.long, device=dev)
        eng.model.set_compaction(True, 8)
        kw = dict(emo_vec=emo_vec, campplus_embedding=style, max_generate_length=560, do_sample=True, num_beams=1, **sample)

        def drained():
            return [eng.model.inference_speech(None, text[i:i + 64], langs=langs[i:i + 64], row_max_new=caps[i:i + 64], **kw)[0] for i in range(0, n_utt, 64)]

        def inflight():
            return eng.model.inference_speech_inflight(None, text, langs=langs, slots=64, chunk_tokens=64, min_free=8, row_max_new=caps, **kw)[0]

        def wall(f):
            best, res = None, None
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = f()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            return best, res
        t_d, c_d = wall(drained)
        t_i, c_i = wall(inflight)
        st = dict(eng.model.last_inflight)

        def lens_of(c):
            return [int((r == eng.model.stop_mel_token).nonzero()[0]) if bool((r == eng.model.stop_mel_token).any()) else int(r.numel()) for r in c]
        ok = [n for c in c_d for n in lens_of(c)] == caps and lens_of(c_i) == caps
        out["inflight_b256"] = {"utterances": n_utt, "slots": 64, "text_tokens": 128, "code_lengths": "uniform 120..560 (seeded), mean %.0f" % (sum(caps) / n_utt),
                                "tokens": int(sum(caps)), "drained_four_batches_s": round(t_d, 4), "inflight_s": round(t_i, 4),
                                "drained_tokens_per_s": round(sum(caps) / t_d, 1), "inflight_tokens_per_s": round(sum(caps) / t_i, 1),
                                "speedup": round(t_d / t_i, 3), "schedule": st, "lengths_as_capped": bool(ok),
                                "what": "GPT stage only (prefill + decode + admissions, host wall time), sampling.  The synthetic weights never emit the stop token "
                                        "(EOS suppressed), so the ragged lengths are GIVEN as per-utterance caps (row_max_new) in both legs; admission itself "
                                        "needs no caps: every slot has its own cache position, step and max_mel_tokens budget"}
    except Exception as e:
        out["inflight_b256"] = {"error": repr(e)}
    try:                                                           # configs[0]: IndexTTS-1.5, one utterance, greedy
        out["config0_v15_single"] = config0_leg(args, dev)
    except Exception as e:
        out["config0_v15_single"] = {"error": repr(e)}
    try:                                                           # configs[3]: IndexTTS-2
        cfg2 = dict(synth.GPT_V2)
        sd2 = dict(eng.gsd)
        sd2.update(synth.cond_weights(cfg2))
        for k in ("spk_emb_proj.weight", "spk_emb_proj.bias", "lang_embedding.weight"):
            sd2.pop(k, None)
        m2 = gpt.UnifiedVoice(**cfg2, precision=args.precision, device=str(dev))
        m2.load_state_dict(sd2)
        m2.post_init_gpt2_config(kv_cache=True, half=args.precision == "bf16")
        s2v2 = s2mel.MyModel(eng.s2_args, use_gpt_latent=True, precision=args.s2mel_precision, device=dev)
        s2v2.load_state_dict({"cfm": synth.s2mel_weights(seed=1234), "length_regulator": synth.regulator_weights(seed=1234),
                              "gpt_layer": synth.gpt_layer_weights()})
        tts2 = infer_v2.IndexTTS2(cfg={"gpt": {"stop_mel_token": 8193}, "version": 2.0}, device=str(dev), frontend=NoFrontend(), gpt=m2,
                                  bigvgan=eng.voc, semantic_codec=eng.codec, s2mel=s2v2, codes_to_mel="engine")
        g = torch.Generator().manual_seed(304)
        spk_feat = torch.randn(1, 750, 1024, generator=g).to(dev)                 # 15 s of w2v-bert features (50 / s)
        emo_feat = torch.randn(1, 750, 1024, generator=g).to(dev)
        b2 = dict(bundle0, spk_cond_emb=spk_feat, emo_cond_emb=emo_feat)
        ev = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev = m2.merge_emovec(spk_feat, emo_feat, torch.tensor([750]), torch.tensor([750]), alpha=0.7)
            torch.cuda.synchronize()
            t_emo = time.perf_counter() - t0
        r = timed(tts2, segments(16, 64, 305), b2, ev, 1.0, dict(sample, num_beams=1, max_mel_tokens=700))
        out["config3_v2_emotion_b16"] = dict(r, batch=16, text_tokens=64, gen_tokens=700, cond_tokens=34, mel_frames=int(700 * 1.72),
                                             emotion_path_ms=t_emo * 1e3,
                                             emotion_path_what="merge_emovec: emotion Conformer (4 blocks) + Perceiver over two 750-frame prompts")
        del tts2, m2, s2v2
        torch.cuda.empty_cache()
    except Exception as e:
        out["config3_v2_emotion_b16"] = {"error": repr(e)}
    return out


def shard_leg(args, eng, text, langs, mel, bundle, n_gen, t_mel):
    """The per-rank shard of the metric's N = 2 / 4 / 8 points timed on THIS GPU: BASELINE.json configs[2] puts 64 / N utterances on each rank
    (strong scaling, `dist.shard_utterances`), every rank holds a full replica and nothing crosses GPUs inside the step but the speaker-bundle
    broadcast (<= 9 MB) and the int16 gather -- so N x (shard audio-seconds / shard step time) is the N-GPU value's upper bound, printed as
    `projected_value_n<N>` and LABELLED a projection (the scaling curve itself is the driver's to measure when a node exists; the reference's own
    multi-GPU guidance is independent replicas, backends/trt/README.md:231-232).  Same `step()`, same generation mode (num_beams = 1), same s2mel
    precision as the headline; one warm step, then two timed.  Also: the headline batch with a 15 s speaker prompt (1292 mel frames: the longest
    the reference accepts, infer_v2_5.py:627 -- the headline uses 6 s), one step."""
    cfgs, proj = {}, {}
    n_total = text.shape[0]
    audio_utt = t_mel * HOP / SR
    for n_gpu in (8, 4, 2):
        b = n_total // n_gpu
        if b < 1:
            continue
        tx, lg, ml = text[:b], langs[:b], mel[:b]
        eng.step(tx, lg, ml, bundle, n_gen, False)
        torch.cuda.synchronize()
        gp = gd = 0.0
        n_steps = 2
        t0 = time.perf_counter()
        for _ in range(n_steps):
            eng.step(tx, lg, ml, bundle, n_gen, False)
            lt = eng.model.last_timing
            gp, gd = gp + lt["prefill_ms"], gd + lt["decode_ms"]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_steps
        cfgs[f"shard_n{n_gpu}"] = {"utterances": b, "ms_per_step": dt * 1e3, "audio_seconds_per_sec": b * audio_utt / dt,
                                   "gpt_prefill_ms": gp / n_steps, "gpt_decode_ms": gd / n_steps,
                                   "gpt_decode_ms_per_token": gd / n_steps / max(1, n_gen - 1),
                                   "gpt_share_of_step": (gp + gd) / n_steps / (dt * 1e3),
                                   "what": f"the {b} utterances one rank of the {n_gpu}-GPU run holds, through the headline's step() on one GPU"}
        proj[f"projected_value_n{n_gpu}"] = {"value": n_gpu * b * audio_utt / dt, "unit": "audio-seconds/sec",
                                             "what": f"PROJECTION, not a measurement of {n_gpu} GPUs: {n_gpu} x the measured single-GPU rate of the per-rank "
                                                     f"shard ({b} utterances); upper bound of the strong-scaling point (excludes the bundle broadcast, "
                                                     "the waveform gather and rank skew)"}
    # 15 s speaker prompt
    fp = 1292
    gb = torch.Generator().manual_seed(5)
    b15 = dict(bundle, ref_mel=(torch.randn(1, mel.shape[1], fp, generator=gb) * 2 - 4).to(mel.device),
               prompt_condition=torch.randn(1, fp, 512, generator=gb).to(mel.device))
    eng.step(text[:2], langs[:2], mel[:2], b15, n_gen, False)           # shapes / tables of the longer prompt, small
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(text, langs, mel, b15, n_gen, False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cfgs["prompt_15s_b64"] = {"utterances": n_total, "prompt_frames": fp, "ms_per_step": dt * 1e3, "audio_seconds_per_sec": n_total * audio_utt / dt,
                              "what": "the headline batch with the longest speaker prompt the reference accepts (15 s = 1292 mel frames, "
                                      "infer_v2_5.py:627) instead of 6 s: attention is quadratic in prompt + target frames; one step"}
    return {"configs": cfgs, "projections": proj}


def config0_leg(args, dev):
    """BASELINE.json configs[0] at its stated size through the product's v1 pipeline class (`indextts_amd.infer.IndexTTS.infer`): IndexTTS-1.5, one
    utterance of 32 text tokens, a 3 s reference clip (conditioning mel (1, 100, 282) at 24 kHz / 256), greedy decode (repetition penalty 10) of 96
    speech tokens (EOS suppressed), the teacher-forced latent pass, the v1 vocoder (ECAPA-TDNN speaker embedding of the reference mel on the
    engine, BigVGAN on the 1280-wide GPT latent, 1024 samples per latent frame) -> 98 304 samples at 24 kHz.  Seeded random weights of that
    architecture; the conditioning Conformer / Perceiver is a once-per-speaker prompt stage and is stood in by a fixed latent, like the speaker
    bundle of the other lines.  The reference runs this config on its CPU path; `cpu_baseline` holds the oracle's per-token / per-frame CPU costs."""
    import warnings
    from indextts_amd import bigvgan, gpt, infer as infer_v1, synth
    cfg15 = dict(synth.GPT_V15)
    g15 = gpt.UnifiedVoiceV1(**cfg15, precision=args.precision, device=str(dev))
    g15.load_state_dict(synth.gpt_v1_weights(cfg15, suppress_eos=True))
    g15.post_init_gpt2_config(kv_cache=True, half=args.precision == "bf16")
    h1 = dict(synth.BIGVGAN_V1_24K)
    v1 = bigvgan.BigVGAN(h1, cond_dim=h1["speaker_embedding_dim"], in_channels=h1["gpt_dim"], device=dev)
    v1.load_state_dict(synth.bigvgan_v1_weights(h1))
    v1.to(dev)
    g = torch.Generator().manual_seed(300)

    class Tok:                                                     # token strings "t<id>": the text front end is not part of the timed path
        def tokenize(self, text): return text.split()
        def split_segments(self, tokens, max_text_tokens_per_segment=120, **kw):
            n = int(max_text_tokens_per_segment)
            return [tokens[i:i + n] for i in range(0, len(tokens), n)]
        def convert_tokens_to_ids(self, tokens): return [int(t[1:]) for t in tokens]

    class Front(infer_v1.FrontendV1):
        tokenizer = Tok()
        mel = (torch.randn(1, 100, 282, generator=g) * 2.0 - 4.0)
        latent = (torch.randn(1, 32, cfg15["model_dim"], generator=g) * 0.5).to(dev)
        def cond_mel(self, audio_prompt, truncate_seconds=None): return self.mel
        def conditioning(self, cond_mel, cond_mel_lengths): return self.latent

    tts = infer_v1.IndexTTS(cfg={"gpt": {"stop_mel_token": 8193, "stop_text_token": 1, "start_text_token": 0}, "version": 1.5}, device=str(dev),
                            use_fp16=args.precision == "bf16", frontend=Front(), gpt=g15, bigvgan=v1)
    text = " ".join("t%d" % int(v) for v in torch.randint(2, 12000, (32,), generator=g))
    runs = []
    for _ in range(6):                                             # one warm call, then the median of five (a 0.1 s call: single shots spread by 30 %)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                       # max_mel_tokens overflow: EOS is suppressed
            sr, wav = tts.infer("prompt.wav", text, None, do_sample=False, num_beams=1, repetition_penalty=10.0, max_mel_tokens=96)
        torch.cuda.synchronize()
        runs.append((time.perf_counter() - t0, {k: round(float(v), 4) for k, v in tts.last_timing.items()}))
    timed = sorted(runs[1:], key=lambda r: r[0])
    dt, stage = timed[len(timed) // 2]
    audio = wav.shape[0] / float(sr)
    res = {"ms": dt * 1e3, "audio_seconds": audio, "audio_seconds_per_sec": audio / dt, "rtf_per_stream": dt / audio, "samples": int(wav.shape[0]),
           "sampling_rate": int(sr), "stage_seconds": stage, "ms_all_timed_calls": [round(r[0] * 1e3, 2) for r in runs[1:]],
           "batch": 1, "text_tokens": 32, "gen_tokens": 96, "cond_mel_frames": 282, "decode": "greedy, repetition_penalty 10"}
    del tts, g15, v1
    torch.cuda.empty_cache()
    return res


def alt_precision_leg(args, eng, precision, text, langs, mel, bundle, n_gen, audio_per_step, stages):
    """--alt-steps full steps with the flow-matching stage in the OTHER precision (same GPT, codec, vocoder, inputs), after the headline:
    one untimed warmup, then the timed steps, the first of them with per-launch profiling for its own stage split."""
    s2 = eng.build_s2(precision)
    cfm = s2.models["cfm"]
    acc = eng.new_stage_acc()
    eng.set_profiling(False)
    eng.render(eng.decode(text, langs, bundle, n_gen, False), mel, bundle, n_gen, False, s2=s2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.alt_steps):
        cfm.set_profiling(k == 0)
        eng.render(eng.decode(text, langs, bundle, n_gen, False), mel, bundle, n_gen, k == 0, s2=s2, acc=acc)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    stages["s2mel_" + precision] = s2_stage(acc, 1, args.prompt_frames, precision)
    del s2
    torch.cuda.empty_cache()
    return {"value": audio_per_step * args.alt_steps / el, "ms_per_step": el / args.alt_steps * 1e3, "steps": args.alt_steps}


def s2_stage(t, n_prof, prompt_frames, precision):
    """stage split of the flow-matching stage from the HIP-event records of `n_prof` profiled steps"""
    n = max(1, n_prof)
    # fp32x3: GEMMs and attention on the bf16 pipe at `x3_products` MFMAs per f32-equivalent one -> peak = bf16 peak / products
    x3p = float(X3_PRODUCTS[0])
    peak = PEAK_BF16_MFMA_TFLOPS if precision == "bf16" else (PEAK_BF16_MFMA_TFLOPS / x3p if precision == "fp32x3" else PEAK_F32_MFMA_TFLOPS)
    gemm_tf = t["gemm_flops"] / max(1e-9, t["gemm_ms"] * 1e-3) / 1e12
    attn_tf = t["attention_flops"] / max(1e-9, t["attention_ms"] * 1e-3) / 1e12
    return {"precision": precision, "codec_regulator_ms_per_step": t["codec_regulator_ms"] / n,
            "cfm_ms_per_step": t["cfm_ms"] / n,                 # 25 Euler steps x CFG batch-2 estimator, host prep included
            "cfm_estimator_ms_per_step": t["estimator_ms"] / n, "cfm_gemm_ms_per_step": t["gemm_ms"] / n,
            "cfm_gemm_tflops": gemm_tf, "cfm_gemm_mfma_frac": gemm_tf / peak, "cfm_gemm_launches_per_step": t["launches"] // n,
            "cfm_attention_ms_per_step": t["attention_ms"] / n, "cfm_attention_tflops": attn_tf,
            "cfm_attention_launches_per_step": t.get("attention_launches", 0) // n,
            "cfm_attention_mfma_frac": attn_tf / peak,
            "cfm_elementwise_ms_per_step": (t["estimator_ms"] - t["gemm_ms"] - t["attention_ms"]) / n,
            "mfma_peak_tflops": peak, "x3_products": int(x3p) if precision == "fp32x3" else None, "prompt_frames": prompt_frames, "euler_steps": EULER_STEPS, "cfg_rate": 0.7}


def gpu_report(args, eng, B, n_text, n_gen, t_mel):
    """roofline (dominant kernel) + stage split from rank 0's HIP-event records of the profiled timed steps."""
    prof_acc, gpt_t, gcfg = eng.prof_acc, eng.gpt_t, eng.gcfg
    n_prof = max(1, eng.n_prof)
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
        "profiled_steps": eng.n_prof,
        "gpt_prefill_ms_per_step": gpt_t["prefill_ms"] / args.steps,
        "gpt_prefill_tflops": prefill_tflops,          # whole prefill pass (GEMMs + attention + LayerNorms) per wall time
        "gpt_prefill_mfma_frac": prefill_tflops / (PEAK_BF16_MFMA_TFLOPS if args.precision == "bf16" else PEAK_F32_MFMA_TFLOPS),
        "gpt_decode_ms_per_step": gpt_t["decode_ms"] / args.steps,
        "gpt_decode_ms_per_token": ms_tok,
        "gpt_decode_algorithmic_GBps": bytes_step / (ms_tok * 1e-3) / 1e9,
        "gpt_decode_hbm_frac": bytes_step / (ms_tok * 1e-3) / 1e9 / PEAK_HBM_GBPS,
        "s2mel": None if eng.s2 is None else s2_stage(eng.s2_t, eng.n_prof, args.prompt_frames, args.s2mel_precision),
        "bigvgan_ms_per_step": sum(v["ms"] for v in prof_acc.values()) / n_prof,
        "bigvgan_kernels": {k: dict(ms_per_step=v["ms"] / n_prof, launches_per_step=v["launches"] // n_prof,
                                    tflops=(v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0),
                                    GBps=(v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0))
                            for k, v in prof_acc.items()},
    }
    x3v = getattr(args, "bigvgan_conv", "f32") == "bf16x3"
    conv_peak = PEAK_BF16_MFMA_TFLOPS / 6.0 if x3v else PEAK_F32_MFMA_TFLOPS
    conv_roof = {"bound": "mfma",
                 "kernel": ("BigVGAN Conv1d's: conv_x3w_kernel + split_tm3_kernel on the >= 96-channel resblocks (three bf16 planes per f32 operand, 6 x "
                            "v_mfma_f32_16x16x32_bf16 per f32-equivalent MFMA; peak = bf16 peak / 6, the split pass's time is inside `achieved`) and "
                            "conv_mfma_kernel (v_mfma_f32_32x32x2_f32) on conv_pre / the upsamplers / the 48- and 24-channel stages"
                            if x3v else "conv_mfma_kernel (BigVGAN Conv1d implicit GEMM, v_mfma_f32_32x32x2_f32)"),
                 "achieved": achieved, "peak": conv_peak, "unit": "TFLOP/s" + (" (f32-equivalent)" if x3v else ""),
                 "frac": achieved / conv_peak, "traffic": traffic if not x3v else None, "traffic_unit": "bytes/launch",
                 "traffic_source": traffic_src, "algorithmic_bytes_per_launch": conv["bytes"] / max(1, conv["launches"]),
                 "launches_per_step": conv["launches"] // n_prof, "avg_launch_ms": conv["ms"] / max(1, conv["launches"]),
                 "ms_per_step": conv["ms"] / n_prof}
    roofline, other = conv_roof, None
    st = stages["s2mel"]
    if st is not None and st["cfm_gemm_ms_per_step"] > conv_roof["ms_per_step"]:
        # the flow-matching GEMMs are the step's dominant kernel (always so with the f32 CFM): all epilogue instantiations of the tile
        # kernel of that precision, algorithmic FLOPs 2 M N K per launch, durations from the HIP events around every launch
        f32 = args.s2mel_precision == "fp32"
        # traffic: PMC passes of the same solve (tools/pmc_s2mel_traffic.sh -> profiles/s2mel_gemm_traffic.json), if the shape matches
        tr, tr_src, alg = None, None, None
        tp2 = os.path.join(ROOT, "profiles", "s2mel_gemm_traffic.json")
        if os.path.exists(tp2):
            tj = json.load(open(tp2))
            if tj.get("B") == B and tj.get("mel_frames") == t_mel and tj.get("precision") == args.s2mel_precision:
                tr, tr_src, alg = tj["hbm_bytes_per_gemm_launch"], tj.get("source"), tj.get("algorithmic_bytes_per_gemm_launch")
        gemm_roof = {"bound": "mfma",
                     "kernel": ("gemm_prefill_kernel<EPI, CONV, VEC, F32 = true> (s2mel DiT / WaveNet GEMMs, v_mfma_f32_16x16x4_f32, every "
                                "epilogue instantiation)" if f32 else
                                f"gemm_x3w8_kernel<EPI, CONV> / gemm_x3_kernel<EPI, CONV, {X3_PRODUCTS[0]}> (s2mel DiT / WaveNet GEMMs with f32 operands as three bf16 planes, "
                                f"{X3_PRODUCTS[0]} x v_mfma_f32_16x16x32_bf16 per f32-equivalent MFMA; achieved / peak in f32-equivalent TFLOP/s, "
                                f"peak = bf16 peak / {X3_PRODUCTS[0]})"
                                if args.s2mel_precision == "fp32x3" else
                                "gemm_tile256_kernel / gemm_prefill_kernel (s2mel DiT / WaveNet GEMMs, v_mfma_f32_16x16x32_bf16, every "
                                "epilogue instantiation)"),
                     "achieved": st["cfm_gemm_tflops"], "peak": st["mfma_peak_tflops"], "unit": "TFLOP/s", "frac": st["cfm_gemm_mfma_frac"],
                     "traffic": tr, "traffic_unit": "bytes/launch", "traffic_source": tr_src, "algorithmic_bytes_per_launch": alg,
                     "launches_per_step": st["cfm_gemm_launches_per_step"],
                     "avg_launch_ms": st["cfm_gemm_ms_per_step"] / max(1, st["cfm_gemm_launches_per_step"]),
                     "ms_per_step": st["cfm_gemm_ms_per_step"]}
        roofline, other = gemm_roof, conv_roof
    out = {"roofline": roofline, "stages": stages}
    if other is not None:
        out["roofline_second_kernel"] = other
    return out


if __name__ == "__main__":
    main()
