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
    for name in ("torchaudio", "torchaudio.functional", "torchaudio.transforms", "torchaudio.compliance",
                 "torchaudio.compliance.kaldi", "librosa", "librosa.util", "librosa.filters"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    if "munch" not in sys.modules:
        m = types.ModuleType("munch")
        m.Munch = Munch
        m.munchify = munchify
        sys.modules["munch"] = m
