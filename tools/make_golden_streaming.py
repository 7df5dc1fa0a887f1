"""Mint tests/golden/streaming.npz: the reference's OWN `StreamingDecoder.generate` (backends/trt/pipeline/streaming.py, imported
from /root/reference -- it needs only numpy) driven by a scripted GPT engine (chunks with overlap, rows finishing at different
chunks, with and without a closing chunk) and a scripted codes -> audio function.  indextts_amd/streaming.py is tested against it."""
import importlib.util
import os

import numpy as np

REF = "/root/reference/backends/trt/pipeline/streaming.py"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_streaming", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def script(case, rng):
    """A list of (n_codes, is_last, batch_done, code_lens) and the audio every chunk renders to."""
    chunk, ovl = case["chunk"], case["overlap"]
    B = case["B"]
    chunks = []
    for i, (n, last, done, lens) in enumerate(case["chunks"]):
        n_samples = int(n * 1.72) * 256
        audio = [rng.uniform(-1.2, 1.2, size=n_samples).astype(np.float32) for _ in range(B)]
        chunks.append(dict(n=n, last=last, done=done, lens=lens, audio=audio))
    return chunks


CASES = [
    dict(B=2, chunk=10, overlap=3, chunks=[(10, False, [False, False], [10, 10]), (10, False, [True, False], [4, 10]), (6, True, [True, True], [0, 6])]),
    dict(B=1, chunk=8, overlap=2, chunks=[(5, True, [True], [5])]),                                       # a single, last chunk
    dict(B=3, chunk=12, overlap=4, chunks=[(12, False, [False, True, False], [12, 9, 12]), (12, False, [False, True, True], [12, 0, 7])]),   # no closing chunk: tails flushed
    dict(B=1, chunk=6, overlap=1, chunks=[(6, False, [False], [6]), (6, False, [False], [6]), (6, False, [False], [6]), (2, True, [True], [2])]),
]


def main():
    ref = load_ref()
    rng = np.random.default_rng(3)
    out = {}
    for ci, case in enumerate(CASES):
        chunks = script(case, rng)

        class Engine:
            def generate_chunks(self, **kw):
                for c in chunks:
                    yield (np.zeros((case["B"], c["n"]), dtype=np.int64), None, c["last"], c["done"], np.asarray(c["lens"]))

        it = iter(chunks)

        def codes_to_audio(codes, latent, lens, prompt_condition, ref_mel, style, prompt_lens=None):
            return next(it)["audio"]

        dec = ref.StreamingDecoder(Engine(), codes_to_audio, chunk_size=case["chunk"], overlap_size=case["overlap"])
        res = list(dec.generate(np.zeros((case["B"], 1, 1)), None, 0, None, None, None))
        out[f"c{ci}_n_yield"] = np.asarray(len(res))
        for yi, (sr, audio, done) in enumerate(res):
            out[f"c{ci}_y{yi}_done"] = np.asarray(done)
            for b, a in enumerate(audio):
                out[f"c{ci}_y{yi}_has{b}"] = np.asarray(a is not None)
                if a is not None:
                    out[f"c{ci}_y{yi}_a{b}"] = a
        for k, c in enumerate(chunks):
            for b in range(case["B"]):
                out[f"c{ci}_in{k}_a{b}"] = c["audio"][b]
    np.savez_compressed(os.path.join(GOLD, "streaming.npz"), **out)
    print("wrote streaming.npz", {k: int(out[k]) for k in out if k.endswith("n_yield")})


if __name__ == "__main__":
    main()
