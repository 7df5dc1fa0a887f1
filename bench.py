#!/usr/bin/env python3
"""Headline benchmark: audio-seconds generated per wall-second, IndexTTS-2.5 hot path, 64-utterance x 128-token batch.

One "step" = one pass of the hot path over one batch of synthetic input on every rank:
    speaker-bundle broadcast (RCCL when N > 1)  ->  GPT speech-token decode (prefill + 560 sampled tokens per utterance,
    top-k 30 / top-p 0.8 / T 0.8 / repetition penalty 10, bf16 weights + KV)  ->  BigVGAN (80-band mel -> 22.05 kHz wave,
    fp32) on a synthetic mel of the length the pipeline would hand over, int(2 * n_tokens * 1.72) frames
    (indextts/infer_v2_5.py:833).  The s2mel stage that sits between the two in the reference runs on PyTorch-ROCm and
    is out of this path, so the mel is synthetic rather than derived from the codes.
Inputs (text ids, conditioning vectors, mel) are resident in HBM before the timed region.  Random-init weights of the
reference architecture (no checkpoints exist offline); EOS is suppressed so every row decodes all 560 tokens.

Contract: `python bench.py --gpus N --steps K --warmup W` (torchrun launches N ranks); rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
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
    return max(1, min(n, 64))


def cpu_baseline(gpt_sd, gpt_cfg, bv_sd, bv_h, n_text, n_gen, t_mel):
    """Reference CPU path restated (oracle/), timed on this box's host cores on a bounded sample.

    GPT: 1 utterance, `n_text` text tokens, prefill + 120 greedy decode steps (fp32, kv-cache) on the full-size stack.
    BigVGAN: 1 utterance x 480 mel frames (fp32).  About 15-25 s of CPU work; scaled to audio-seconds/second for an
    utterance of `n_gen` tokens / `t_mel` frames (decode cost per token grows with context; the sample covers the first
    120 of `n_gen` positions, which favours the CPU).
    """
    from oracle import bigvgan_oracle as BO
    from oracle import gpt_oracle as GO
    cores = min(usable_cores(), max(1, torch.get_num_threads()))     # never above torch's own default
    torch.set_num_threads(cores)
    log(f"[bench] cpu_baseline on {cores} threads (os.cpu_count()={os.cpu_count()})")
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
    audio_s = t_mel * HOP / SR
    cpu_time = t_prefill + n_gen * t_tok + t_mel * t_frame
    return {"value": audio_s / cpu_time, "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
            "sample": f"oracle (torch fp32 CPU restatement): GPT 1 utt x {n_text} text tokens, prefill + {steps} greedy "
                      f"decode steps; BigVGAN 1 utt x {frames} mel frames; extrapolated to {n_gen} tokens / {t_mel} frames",
            "gpt_ms_per_token": t_tok * 1e3, "gpt_prefill_ms": t_prefill * 1e3, "bigvgan_ms_per_frame": t_frame * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utts", type=int, default=64, help="utterances per GPU (weak scaling: global batch = utts * N)")
    ap.add_argument("--text-tokens", type=int, default=128)
    ap.add_argument("--gen-tokens", type=int, default=560)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--bigvgan-chunk", type=int, default=0, help="utterances per BigVGAN launch group (0 = all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON of rank 0): native libraries (RCCL prints a version banner on init) and any
    # stray print write to fd 1, so fd 1 is pointed at stderr for the run and the result goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("ITTS_BENCH_FORCE_DIST") == "1":      # the env switch exercises the RCCL path on one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from indextts_amd import bigvgan, gpt, synth
    t_load = time.perf_counter()
    gcfg = dict(synth.GPT_V25)
    gsd = synth.gpt_weights(gcfg, seed=1234, suppress_eos=True)
    model = gpt.UnifiedVoice(**gcfg, spk_cond_mode="campplus", precision=args.precision, device=str(dev))
    model.load_state_dict(gsd)
    model.post_init_gpt2_config(kv_cache=True, half=args.precision == "bf16")
    model.use_graph = not args.no_graph
    bh = dict(synth.BIGVGAN_V2_22K)
    bsd = synth.bigvgan_weights(bh, seed=1234)
    voc = bigvgan.BigVGAN(bh)
    voc.load_state_dict(bsd)
    voc.to(dev)
    voc.set_profiling(True)
    if rank == 0:
        log(f"[bench] weights synthesised + packed + uploaded in {time.perf_counter() - t_load:.1f}s")

    B, n_text, n_gen = args.utts, args.text_tokens, args.gen_tokens
    t_mel = int(2 * n_gen * 1.72)
    g = torch.Generator().manual_seed(100 + rank)
    text = torch.randint(2, gcfg["number_text_tokens"], (B, n_text), generator=g).to(dev)
    langs = torch.full((B,), 3, dtype=torch.long, device=dev)
    mel = (torch.randn(B, bh["num_mels"], t_mel, generator=g) * 2 - 4).to(dev)
    # speaker bundle: produced by the prompt encoders on rank 0 in the real pipeline, broadcast once per batch
    style = torch.randn(1, 192, generator=torch.Generator().manual_seed(5)).to(dev)
    emo_vec = (torch.randn(1, gcfg["model_dim"], generator=torch.Generator().manual_seed(6)) * 0.1).to(dev)
    gen_kw = dict(do_sample=True, top_p=0.8, top_k=30, temperature=0.8, num_beams=1, repetition_penalty=10.0,
                  length_penalty=0.0)

    prof_acc = {}
    gpt_t = {"prefill_ms": 0.0, "decode_ms": 0.0, "steps": 0}

    def one_step(record):
        bundle_s, bundle_e = style, emo_vec
        if dist is not None:
            if rank != 0:
                bundle_s, bundle_e = torch.empty_like(style), torch.empty_like(emo_vec)
            dist.broadcast(bundle_s, 0)
            dist.broadcast(bundle_e, 0)
        codes, _ = model.inference_speech(None, text, langs=langs, emo_vec=bundle_e, campplus_embedding=bundle_s,
                                          max_generate_length=n_gen, **gen_kw)
        assert codes.shape == (B, n_gen), codes.shape
        chunk = args.bigvgan_chunk or B
        outs = []
        for b0 in range(0, B, chunk):
            outs.append(voc(mel[b0:b0 + chunk]))
            if record:
                for k, v in voc.profile().items():
                    a = prof_acc.setdefault(k, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
                    for kk in a:
                        a[kk] += v[kk]
        if record:
            for k in ("prefill_ms", "decode_ms", "steps"):
                gpt_t[k] += model.last_timing[k]
        return codes, outs

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        tw = time.perf_counter()
        one_step(False)
        torch.cuda.synchronize()
        if rank == 0:
            log(f"[bench] warmup step {i}: {time.perf_counter() - tw:.2f}s")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, wavs = one_step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    audio_per_step = world * B * (t_mel * HOP) / SR
    value = audio_per_step * args.steps / elapsed
    if rank == 0:
        conv = prof_acc.get("conv1d_mfma", dict(ms=1e-9, launches=1, flops=0.0, bytes=0.0))
        achieved = conv["flops"] / (conv["ms"] * 1e-3) / 1e12
        # HBM traffic of the same kernel: PMC counters cannot be read from inside this process, so the figure comes from
        # the committed rocprofv3 --pmc summary of the same forward (tools/pmc_bench_traffic.sh), if it matches this shape.
        traffic, traffic_src = None, None
        tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "conv_traffic.json")
        if os.path.exists(tp):
            tj = json.load(open(tp))
            if tj.get("B") == B and tj.get("mel_frames") == t_mel:
                traffic, traffic_src = tj["hbm_bytes_per_conv_dispatch"], "profiles/conv_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
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
            "bigvgan_ms_per_step": sum(v["ms"] for v in prof_acc.values()) / args.steps,
            "bigvgan_kernels": {k: dict(ms_per_step=v["ms"] / args.steps, launches_per_step=v["launches"] // args.steps,
                                        tflops=(v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0),
                                        GBps=(v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0))
                                for k, v in prof_acc.items()},
        }
        out = {
            "metric": "audio-seconds/sec (RTF) IndexTTS-2.5, 64-utt batch @1/2/4/8 MI355X",
            "value": value, "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 (GPT weights/KV/GEMM inputs, f32 accumulate) + f32 (BigVGAN)" if args.precision == "bf16" else "f32",
            "data": "synthetic (seeded random-init weights of the IndexTTS-2.5 architecture; synthetic text ids, "
                    "conditioning vectors and mel; EOS suppressed so every row decodes all tokens)",
            "rtf": 1.0 / value * world * B,          # wall seconds per audio second of ONE utterance stream
            "config": {"workload": f"IndexTTS-2.5 hot path, {B} utterances/GPU x {n_text} text tokens -> {n_gen} speech "
                                   f"tokens (top-k 30, top-p 0.8, T 0.8, rep-penalty 10, num_beams 1) + BigVGAN-v2 22 kHz "
                                   f"on {t_mel}-frame mels (BASELINE.json configs[2] shape, {B} utterances per GPU)",
                       "global_batch": world * B, "text_tokens": n_text, "gen_tokens": n_gen, "mel_frames": t_mel,
                       "audio_seconds_per_utt": t_mel * HOP / SR, "parallelism": f"utterance-dp{world}",
                       "use_hipgraph": not args.no_graph},
            "roofline": {"bound": "mfma", "kernel": "conv_mfma_kernel (BigVGAN Conv1d implicit GEMM, v_mfma_f32_32x32x2_f32)",
                         "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": conv["bytes"] / max(1, conv["launches"]),
                         "launches_per_step": conv["launches"] // max(1, args.steps),
                         "avg_launch_ms": conv["ms"] / max(1, conv["launches"])},
            "stages": stages,
        }
        log("[bench] GPU result:", json.dumps({k: out[k] for k in ("value", "ms_per_step", "roofline", "stages")}))
        if not args.no_cpu_baseline and world == 1:
            t_cpu = time.perf_counter()
            try:
                out["cpu_baseline"] = cpu_baseline(gsd, gcfg, bsd, bh, n_text, n_gen, t_mel)
            except Exception as e:      # never lose the GPU line because the baseline leg failed
                out["cpu_baseline"] = {"error": repr(e)}
            log(f"[bench] cpu_baseline leg took {time.perf_counter() - t_cpu:.1f}s")
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
