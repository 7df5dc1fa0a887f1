"""GPU box: is the s2mel solve deterministic run to run?  Repeats one CFG Euler solve (and one estimator call) per mode on the same inputs and
prints a hash of the output bits.  usage: s2mel_determinism.py [n_utts] [prompt] [gen] [steps] [reps] spec ...   (spec as in s2mel_bench.py)"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from indextts_amd import _lib, s2mel, synth  # noqa: E402

B, Tp, Tg, steps, reps = (int(v) for v in sys.argv[1:6])
specs = sys.argv[6:]
import copy
args = copy.deepcopy(synth.S2MEL_V2)
args["DiT"]["depth"] = int(os.environ.get("DEPTH", args["DiT"]["depth"]))
args["wavenet"]["num_layers"] = int(os.environ.get("WN_LAYERS", args["wavenet"]["num_layers"]))
SOLVE = os.environ.get("SOLVE", "1") != "0"
POISON = os.environ.get("POISON")          # byte value the cached workspace is filled with before every call (0xff: NaN patterns)
g = torch.Generator().manual_seed(0)
T = Tp + Tg
x = torch.randn(B, 80, T, generator=g).cuda()
mu = torch.randn(B, T, args["DiT"]["content_dim"], generator=g).cuda()
prompt = (torch.randn(1, 80, Tp, generator=g) * 0.5 - 1.0).cuda()
style = torch.randn(1, args["style_encoder"]["dim"], generator=g).cuda()
t_span = torch.linspace(0, 1, steps + 1)
lens = torch.full((B,), T)
hsh = lambda y: hashlib.sha1(y.float().cpu().numpy().tobytes()).hexdigest()[:10]
for spec in specs:
    prec, _, optstr = spec.partition(":")
    opts = {k: int(v) for k, v in (kv.split("=") for kv in optstr.split(",") if kv)}
    prune = opts.pop("prune", 1)
    with _lib.option_scope(**opts):
        m = s2mel.CFM(args, precision=prec, device="cuda:0")
        m.load_state_dict(synth.s2mel_weights(args, seed=1234))
        m.prune_dead_rows = bool(prune)
        ys = [m.solve_euler(x.clone(), lens, prompt, mu, style, None, t_span, 0.7, frame_lens=[T] * B) if SOLVE else torch.zeros(1) for _ in range(reps)]
        torch.cuda.synchronize()
        px = torch.zeros_like(x); px[..., :Tp] = prompt
        def poison():
            if POISON is not None and m._ws is not None:
                m._ws.fill_(int(POISON, 0))
        es = []
        for _ in range(reps):
            poison()
            es.append(m.estimator(torch.cat([x, x]), torch.cat([px, torch.zeros_like(px)]), lens, torch.full((2 * B,), 0.3),
                          torch.cat([style.expand(B, -1), torch.zeros(B, style.shape[1], device="cuda")]), torch.cat([mu, torch.zeros_like(mu)])))
        torch.cuda.synchronize()
    del m
    ds = [float((y - ys[0]).abs().max()) for y in ys[1:]]
    de = [float((e - es[0]).abs().max()) for e in es[1:]]
    nbad = [int(((e - es[0]).abs() > 1e-4).sum()) for e in es[1:]]
    print(f"poison {POISON} finite {[bool(torch.isfinite(e).all()) for e in es]} depth {args['DiT']['depth']} wavenet {args['wavenet']['num_layers']} {spec}: solve hashes {[hsh(y) for y in ys]} max|d| vs run 0 {ds}; estimator hashes {[hsh(e) for e in es]} max|d| {de} elements off by > 1e-4: {nbad}", flush=True)
    if nbad and max(nbad):
        bad = ((es[1] - es[0]).abs() > 1e-4).nonzero()
        print("   first bad (utt, channel, frame):", bad[:12].tolist(), flush=True)
