"""GPU box: WHAT differs when a stage of the s2mel estimator is not bit-stable run to run?  tools/s2mel_trace.py names the first differing stage
from the engine's checksums; this tool goes one step further with itts_s2mel_set_capture: pass 1 finds the most frequent first-differing stage
label, pass 2 repeats the call with a copy of every output of the stages carrying that label, and prints for every repetition that differs from
repetition 0 which elements differ (row / column pattern, reference and observed values, whether the observed value is the reference value of
another place: the same row of the previous layer's image (a store that never landed), a neighbouring row (wrong row metadata), ...).
usage: s2mel_capture.py n_utts prompt gen reps spec ...   (spec = precision[:option=value,...]; env DEPTH / WN_LAYERS / LABEL to force the label)"""
import collections
import copy
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import _lib, s2mel, synth  # noqa: E402

B, Tp, Tg, reps = (int(v) for v in sys.argv[1:5])
specs = sys.argv[5:]
args = copy.deepcopy(synth.S2MEL_V2)
args["DiT"]["depth"] = int(os.environ.get("DEPTH", args["DiT"]["depth"]))
args["wavenet"]["num_layers"] = int(os.environ.get("WN_LAYERS", args["wavenet"]["num_layers"]))
H = args["DiT"]["hidden_dim"]
g = torch.Generator().manual_seed(0)
T = Tp + Tg
x = torch.randn(B, 80, T, generator=g).cuda()
mu = torch.randn(B, T, args["DiT"]["content_dim"], generator=g).cuda()
prompt = (torch.randn(1, 80, Tp, generator=g) * 0.5 - 1.0).cuda()
style = torch.randn(1, args["style_encoder"]["dim"], generator=g).cuda()
lens = torch.full((B,), T)
px = torch.zeros_like(x)
px[..., :Tp] = prompt
CAP = 4096
CAP_BYTES = 3 << 30
L = _lib.lib()
N = 2 * B * T
t_pad = (T + 63) // 64 * 64


def call(m):
    m.estimator(torch.cat([x, x]), torch.cat([px, torch.zeros_like(px)]), lens, torch.full((2 * B,), 0.3),
                torch.cat([style.expand(B, -1), torch.zeros(B, style.shape[1], device="cuda")]), torch.cat([mu, torch.zeros_like(mu)]))
    torch.cuda.synchronize()


def width_of(label, nbytes, esz):
    """row width (elements) used to print (row, column) of a flat element index"""
    if label.endswith("-> K"):
        return 64
    if label.endswith("-> V^T"):
        return t_pad
    n = nbytes // esz
    return n // N if n % N == 0 else H


def describe(label, ref, got, prev_ref, esz, prec):
    dt = torch.bfloat16 if esz == 2 else torch.float32
    it = torch.int16 if esz == 2 else torch.int32
    r, o = ref.view(it), got.view(it)
    bad = (r != o).nonzero().flatten()
    W = width_of(label, ref.numel(), esz)
    rows = torch.unique(bad // W)
    cols = torch.unique(bad % W)
    print(f"      {bad.numel()} of {r.numel()} elements differ; {rows.numel()} rows (width {W}), {cols.numel()} distinct columns; "
          f"rows {rows[:24].tolist()}{' ...' if rows.numel() > 24 else ''}")
    print(f"      columns {cols[:40].tolist()}{' ...' if cols.numel() > 40 else ''}")
    rf, of = ref.view(dt).float(), got.view(dt).float()
    d = (rf - of).abs()
    print(f"      max |ref - got| {float(d.max()):.4e}; ref rms {float(rf.pow(2).mean().sqrt()):.4e}")
    for row in rows[:6].tolist():
        seg = slice(row * W, (row + 1) * W)
        bcols = (r[seg] != o[seg]).nonzero().flatten()
        c0 = int(bcols[0])
        print(f"      row {row}: {bcols.numel()} of {W} columns differ (first {bcols[:8].tolist()}); ref[{c0}..] {rf[seg][c0:c0 + 6].tolist()} got {of[seg][c0:c0 + 6].tolist()}")
        if label.endswith("-> K"):
            sh, t = divmod(row, t_pad)
            print(f"         = (sequence * heads + head) {sh}, frame {t}")
            for dt_ in (-2, -1, 1, 2):                       # the same head's neighbouring frames: wrong row metadata would put another frame's rotation here
                rr = row + dt_
                if 0 <= rr < r.numel() // W and bool((r[rr * W:(rr + 1) * W] == o[seg]).all()):
                    print(f"         got == ref of frame {t + dt_}")
        if bool((o[seg] == 0).all()):
            print("         got is all zero (the store never landed on a cleared buffer)")
        if prev_ref is not None and bool((prev_ref.view(it)[seg] == o[seg]).all()):
            print("         got == the previous layer's image of this row (the store never landed)")
    # 128-byte line pattern of the differing bytes
    lines = torch.unique((bad * esz) // 128)
    print(f"      {lines.numel()} distinct 128-byte lines; elements per touched line: {bad.numel() / max(1, lines.numel()):.1f} of {128 // esz}")


for spec in specs:
    prec, _, optstr = spec.partition(":")
    opts = {k: int(v) for k, v in (kv.split("=") for kv in optstr.split(",") if kv)}
    esz = 2 if prec == "bf16" else 4
    with _lib.option_scope(**opts):
        m = s2mel.CFM(args, precision=prec, device="cuda:0")
        m.load_state_dict(synth.s2mel_weights(args, seed=1234))
        buf = torch.zeros(CAP, dtype=torch.int64, device="cuda:0")
        _lib.check(L.itts_s2mel_set_trace(m._h, _lib.ptr(buf), CAP), "itts_s2mel_set_trace")
        # pass 1: which label differs first, most often
        runs = []
        for _ in range(reps):
            call(m)
            n = L.itts_s2mel_trace_count(m._h)
            runs.append(tuple(buf[:n].cpu().tolist()))
        labels = [(L.itts_s2mel_trace_label(m._h, i) or b"?").decode() for i in range(len(runs[0]))]
        major, cnt = collections.Counter(runs).most_common(1)[0]
        firsts = collections.Counter()
        for run in runs:
            if run != major:
                diff = [i for i in range(len(major)) if run[i] != major[i]]
                firsts[labels[diff[0]]] += 1
        print(f"{spec} depth {args['DiT']['depth']} wavenet {args['wavenet']['num_layers']} B={B} T={T}: pass 1: {cnt} of {reps} repetitions agree; "
              f"first differing labels {dict(firsts)}", flush=True)
        label = os.environ.get("LABEL") or (firsts.most_common(1)[0][0] if firsts else None)
        if label is None:
            del m
            continue
        # pass 2: images of every stage with that label; repetition 0 of the pass whose checksums equal the majority is the reference
        cap_a = torch.empty(CAP_BYTES, dtype=torch.uint8, device="cuda:0")
        cap_b = torch.empty(CAP_BYTES, dtype=torch.uint8, device="cuda:0")
        ref_run = None
        shown = 0
        for rep in range(3 * reps):
            tgt = cap_a if ref_run is None else cap_b
            prefix = "wqkv" if label.startswith("wqkv") else label      # Q, K and V^T of every layer when the wqkv GEMM is the suspect
            _lib.check(L.itts_s2mel_set_capture(m._h, _lib.ptr(tgt), CAP_BYTES, prefix.encode()), "itts_s2mel_set_capture")
            call(m)
            n = L.itts_s2mel_trace_count(m._h)
            run = tuple(buf[:n].cpu().tolist())
            if ref_run is None:
                if run == major:
                    ref_run = run
                continue
            if run == major:
                continue
            diff = [i for i in range(len(major)) if run[i] != major[i]]
            i0 = diff[0]
            nb = C.c_size_t(0)
            off = L.itts_s2mel_capture_offset(m._h, i0, C.byref(nb))
            print(f"   pass 2 repetition {rep}: first differing stage {i0} = '{labels[i0]}' ({len(diff)} stages differ); captured at {off} ({nb.value} bytes)", flush=True)
            if off < 0:
                continue
            prev = None
            for j in range(i0 - 1, -1, -1):                 # the previous layer's image of the same label (what the buffer held before this launch)
                if labels[j] == labels[i0]:
                    nb2 = C.c_size_t(0)
                    o2 = L.itts_s2mel_capture_offset(m._h, j, C.byref(nb2))
                    if o2 >= 0 and nb2.value == nb.value:
                        prev = cap_a[o2:o2 + nb2.value]
                    break
            describe(labels[i0], cap_a[off:off + nb.value], cap_b[off:off + nb.value], prev, esz, prec)
            shown += 1
            if shown >= 6:
                break
        if ref_run is None:
            print("   pass 2: no repetition matched the majority checksums", flush=True)
        _lib.check(L.itts_s2mel_set_capture(m._h, None, 0, None), "itts_s2mel_set_capture")
        _lib.check(L.itts_s2mel_set_trace(m._h, None, 0), "itts_s2mel_set_trace")
        del cap_a, cap_b
    del m
