"""Mint tests/golden/host_v1.json: inputs/outputs of the reference's OWN v1 pipeline helpers
(`IndexTTS.remove_long_silence`, `bucket_segments`, `pad_tokens_cat`, indextts/infer.py:135-268), AST-extracted from
/root/reference and run here on seeded inputs.  The host mirror (indextts_amd/infer.py) is tested against this file."""
import ast
import json
import os
import random
import textwrap
from typing import Dict, List  # noqa: F401  (names used by the extracted source)

import torch
from torch.nn.utils.rnn import pad_sequence  # noqa: F401

REF = "/root/reference/indextts/infer.py"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def extract(names):
    src = open(REF).read()
    out = {}
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == "IndexTTS":
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name in names:
                    out[sub.name] = textwrap.dedent(ast.get_source_segment(src, sub, padded=True))
    return out


class Cfg:
    class gpt:
        stop_text_token = 1
        start_text_token = 0


def main():
    ns = dict(torch=torch, pad_sequence=pad_sequence, List=List, Dict=Dict)
    pieces = extract({"remove_long_silence", "bucket_segments", "pad_tokens_cat"})

    class Ref:
        stop_mel_token = 8193
        cfg = Cfg
        model_version = None

    for k, v in pieces.items():
        exec(compile(v, REF + ":" + k, "exec"), ns)
        setattr(Ref, k, ns[k])
    rnd = random.Random(7)
    out = {"silence": [], "buckets": [], "pad": []}
    for case in range(12):
        B = rnd.choice([1, 1, 2, 3])
        T = rnd.choice([20, 60, 90])
        rows = []
        for b in range(B):
            row = []
            while len(row) < T:
                if rnd.random() < 0.35:
                    row += [52] * rnd.choice([1, 3, 9, 10, 11, 15, 25, 40])
                else:
                    row += [rnd.randrange(100, 8000) for _ in range(rnd.choice([1, 2, 5]))]
            row = row[:T]
            if rnd.random() < 0.7:
                e = rnd.randrange(T // 2, T)
                row[e:] = [8193] * (T - e)
            rows.append(row)
        codes = torch.tensor(rows, dtype=torch.long)
        r = Ref()
        oc, ol = r.remove_long_silence(codes.clone(), silent_token=52, max_consecutive=30)
        out["silence"].append({"codes": rows, "out": oc.tolist(), "lens": ol.tolist()})
    for case in range(10):
        n = rnd.choice([3, 5, 8, 13, 20])
        segs = [["t"] * rnd.choice([0, 2, 5, 9, 10, 14, 22, 30, 31, 60, 90]) for _ in range(n)]
        for bm in (1, 2, 4):
            r = Ref()
            res = r.bucket_segments(segs, bucket_max_size=bm)
            out["buckets"].append({"lens": [len(s) for s in segs], "bucket_max_size": bm,
                                   "out": [[it["idx"] for it in b] for b in res]})
    for ver in (None, 1.5):
        for case in range(4):
            toks = [torch.randint(2, 100, (1, rnd.choice([3, 5, 9, 20])), generator=torch.Generator().manual_seed(case * 7 + i))
                    for i in range(rnd.choice([2, 3, 4]))]
            r = Ref()
            r.model_version = ver
            try:
                res = r.pad_tokens_cat([t.clone() for t in toks])
            except TypeError as e:           # torch < 2.6: pad_sequence has no padding_side
                print("pad_tokens_cat v1.5 not runnable on this torch:", e)
                continue
            out["pad"].append({"version": ver, "tokens": [t.tolist() for t in toks], "out": res.tolist()})
    with open(os.path.join(GOLD, "host_v1.json"), "w") as f:
        json.dump(out, f)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
