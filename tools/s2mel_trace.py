"""GPU box: which stage of the s2mel estimator is not bit-stable run to run?  Repeats one CFG-stacked estimator call on the same inputs with the
engine's trace checksums on (itts_s2mel_set_trace: one order-independent 64-bit checksum per stage output, in launch order) and prints, for every
repetition that differs from the majority, the FIRST stage whose output differs.
usage: s2mel_trace.py [n_utts] [prompt] [gen] [reps] spec ...   (spec = precision[:option=value,...] as in s2mel_bench.py; env DEPTH / WN_LAYERS)"""
import collections
import copy
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
L = _lib.lib()
for spec in specs:
    prec, _, optstr = spec.partition(":")
    opts = {k: int(v) for k, v in (kv.split("=") for kv in optstr.split(",") if kv)}
    with _lib.option_scope(**opts):
        m = s2mel.CFM(args, precision=prec, device="cuda:0")
        m.load_state_dict(synth.s2mel_weights(args, seed=1234))
        buf = torch.zeros(CAP, dtype=torch.int64, device="cuda:0")
        _lib.check(L.itts_s2mel_set_trace(m._h, _lib.ptr(buf), CAP), "itts_s2mel_set_trace")
        runs = []
        for _ in range(reps):
            buf.zero_()
            m.estimator(torch.cat([x, x]), torch.cat([px, torch.zeros_like(px)]), lens, torch.full((2 * B,), 0.3),
                        torch.cat([style.expand(B, -1), torch.zeros(B, style.shape[1], device="cuda")]), torch.cat([mu, torch.zeros_like(mu)]))
            torch.cuda.synchronize()
            n = L.itts_s2mel_trace_count(m._h)
            runs.append(tuple(buf[:n].cpu().tolist()))
        labels = [(L.itts_s2mel_trace_label(m._h, i) or b"?").decode() for i in range(len(runs[0]))]
        _lib.check(L.itts_s2mel_set_trace(m._h, None, 0), "itts_s2mel_set_trace")
    del m
    major, cnt = collections.Counter(runs).most_common(1)[0]
    print(f"{spec} depth {args['DiT']['depth']} wavenet {args['wavenet']['num_layers']} B={B} T={T}: {len(major)} stage checksums per call; "
          f"{cnt} of {reps} repetitions agree on all of them", flush=True)
    firsts = collections.Counter()
    for r, run in enumerate(runs):
        if run == major:
            continue
        diff = [i for i in range(min(len(run), len(major))) if run[i] != major[i]]
        i0 = diff[0] if diff else -1
        firsts[(i0, labels[i0] if i0 >= 0 else "length")] += 1
        print(f"   repetition {r}: first differing stage {i0} = '{labels[i0] if i0 >= 0 else '?'}' ({len(diff)} of {len(major)} stages differ; the stage before: "
              f"'{labels[i0 - 1] if i0 > 0 else '-'}')", flush=True)
    if firsts:
        print("   first-differing-stage histogram:", dict(firsts), flush=True)
