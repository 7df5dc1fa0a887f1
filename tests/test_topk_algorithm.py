"""The top-k threshold selection of the sampling kernels, restated lane by lane in numpy (gpt_kernels.hip::topk_kth_key, including its
-DITTS_TOPK_V2 variant: lower bound from the per-lane maxima, compaction of the keys above it, bisection of the survivors), against a sort.
This checks the ALGORITHM (the selection logic a wave executes); the HIP code itself is checked on the GPU by the id-equality tests."""
import numpy as np
import pytest

KPT, LANES, WAVES = 33, 64, 4


def f2key(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint64)


def wave_kth(keys, k, v2):
    """keys: (KPT, LANES) uint64 of one wave (0 = past the row end) -> (T_w, emitted list of exactly k keys)"""
    def bisect(count_ge):
        t = 0
        for bit in range(31, -1, -1):
            c = t | (1 << bit)
            if count_ge(c) >= k:
                t = c
        return t
    T = None
    if v2:
        mx = keys.max(axis=0)                                       # per-lane maxima
        L = bisect(lambda c: int((mx >= c).sum()))
        live = keys[(keys >= L) & (L != 0)]
        if L != 0 and live.size <= 128:
            T = 0
            for bit in range(31, -1, -1):
                c = T | (1 << bit)
                if c <= L or int((live >= c).sum()) >= k:
                    T = c
    if T is None:
        T = bisect(lambda c: int((keys >= c).sum()))
    above = keys[keys > T]
    assert above.size < k or T == 0xFFFFFFFF
    emitted = list(above) + [T] * (k - above.size)
    return T, emitted[:k] if len(emitted) >= k else emitted + [0] * (k - len(emitted))


def block_kth(scores, k, v2):
    V = scores.size
    keys = np.zeros(KPT * 256, dtype=np.uint64)
    keys[:V] = f2key(scores)
    per_thread = keys.reshape(KPT, 256)                              # element tid + 256 j sits in thread tid, register j
    cand = []
    for w in range(WAVES):
        _, em = wave_kth(per_thread[:, 64 * w:64 * (w + 1)], k, v2)
        cand += em
    cand = np.array(cand, dtype=np.uint64)
    g = 0
    for bit in range(31, -1, -1):
        c = g | (1 << bit)
        if int((cand >= c).sum()) >= k:
            g = c
    return g


@pytest.mark.parametrize("v2", [False, True])
@pytest.mark.parametrize("V,k", [(8194, 30), (8194, 1), (8194, 64), (300, 30), (70, 6), (8448, 50)])
def test_selection_finds_the_kth_largest_key(V, k, v2):
    rng = np.random.default_rng(V * 131 + k)
    for trial in range(6):
        x = (rng.standard_normal(V) * 4).astype(np.float32)
        if trial == 1:
            x[rng.integers(0, V, V // 3)] = -np.inf                  # masked entries
        if trial == 2:
            x = np.round(x)                                          # heavy ties
        if trial == 3:
            x[:] = 1.5                                               # everything equal
        if trial == 4:
            x[rng.integers(0, V, 5)] = 1e30                          # a few dominant scores
        want = np.sort(f2key(x))[::-1][k - 1]
        assert block_kth(x, k, v2) == want, (V, k, trial, v2)
