"""Import helpers for running the reference's s2mel classes (indextts/s2mel/modules/*) in this container: stubs for the
packages the reference imports at module level but that the flow-matching / DiT path never calls (torchaudio, librosa,
munch).  Used by tools/make_golden_s2mel.py only."""
import sys
import types

REF = "/root/reference"


class Munch(dict):
    """attribute-access dict (the subset of `munch.Munch` the reference uses for its config objects)"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def munchify(d):
    if isinstance(d, dict):
        return Munch({k: munchify(v) for k, v in d.items()})
    return d


def install():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name in ("torchaudio", "torchaudio.functional", "torchaudio.functional.functional", "torchaudio.transforms",
                 "torchaudio.compliance", "torchaudio.compliance.kaldi", "librosa", "librosa.util", "librosa.filters"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    # names imported at module level by indextts/codec/kmeans/vocos.py (mel-scale helpers of a feature extractor the decode
    # path never builds)
    ff = sys.modules["torchaudio.functional.functional"]
    for fn in ("_hz_to_mel", "_mel_to_hz"):
        if not hasattr(ff, fn):
            setattr(ff, fn, lambda *a, **k: (_ for _ in ()).throw(RuntimeError("torchaudio stub")))
    # indextts/s2mel/modules/length_regulator.py imports VectorQuantize from the vendored DAC package, whose __init__ needs
    # `audiotools`; the regulator only instantiates it when vector_quantize=True (not the v2 / v2.5 configuration)
    for name in ("indextts.s2mel.dac", "indextts.s2mel.dac.nn", "indextts.s2mel.dac.nn.quantize"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    q = sys.modules["indextts.s2mel.dac.nn.quantize"]
    if not hasattr(q, "VectorQuantize"):
        q.VectorQuantize = type("VectorQuantize", (), {"__init__": lambda self, *a, **k: (_ for _ in ()).throw(RuntimeError("dac stub"))})
    if "munch" not in sys.modules:
        m = types.ModuleType("munch")
        m.Munch = Munch
        m.munchify = munchify
        sys.modules["munch"] = m
