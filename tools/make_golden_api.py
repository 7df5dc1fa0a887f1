"""Mint tests/golden/api_signatures.json: the constructor / infer() / infer_generator() signatures of the reference's three pipeline
classes, read from the reference SOURCE by `ast` (no import: the modules pull torchaudio / librosa / modelscope).  Each entry is the
ordered list of (name, default-as-source-text | null) up to but excluding **kwargs, plus the name of the **kwargs parameter.

    python tools/make_golden_api.py          (this container only: reads /root/reference)
"""
import ast
import json
import os

REF = "/root/reference/indextts"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "api_signatures.json")
WANT = {"infer.py": ("IndexTTS", ["__init__", "infer", "infer_fast"]),
        "infer_v2.py": ("IndexTTS2", ["__init__", "infer", "infer_generator"]),
        "infer_v2_5.py": ("IndexTTS2", ["__init__", "infer", "infer_generator"])}


def sig(fn: ast.FunctionDef):
    a = fn.args
    names = [x.arg for x in a.posonlyargs + a.args]
    defaults = [None] * (len(names) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
    return {"params": [[n, d] for n, d in zip(names, defaults)][1:],          # drop self
            "kwonly": [[x.arg, ast.unparse(d) if d is not None else None] for x, d in zip(a.kwonlyargs, a.kw_defaults)],
            "var_kw": a.kwarg.arg if a.kwarg else None}


out = {}
for fname, (cls, methods) in WANT.items():
    tree = ast.parse(open(os.path.join(REF, fname)).read())
    c = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    for m in methods:
        fn = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == m)
        out[f"{fname}:{cls}.{m}"] = dict(sig(fn), line=fn.lineno)
json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
print("wrote", os.path.normpath(OUT), len(out), "signatures")
