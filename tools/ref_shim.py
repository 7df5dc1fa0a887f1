"""Import pieces of the REFERENCE's vendored HF code under the transformers version installed here.

Build-container only (needs /root/reference).  The reference pins transformers==4.52.1; this image has
5.15, where a number of private names the vendored files import no longer exist.  The loop below retries
an import, and for every `cannot import name X from M` / `No module named M` it injects an inert stub
(a dummy class / empty module) -- none of the stubs are on the code paths exercised by the goldens
(beam constraints, quantized caches, candidate generators, ...).  Used by tools/make_golden_gpt.py.
"""
import importlib
import importlib.util
import re
import sys
import types

REF = "/root/reference"


def _stub_module(name: str):
    mod = types.ModuleType(name)
    mod.__stub__ = True
    sys.modules[name] = mod
    parent, _, child = name.rpartition(".")
    if parent:
        if parent not in sys.modules:
            try:
                importlib.import_module(parent)
            except Exception:
                _stub_module(parent)
        setattr(sys.modules[parent], child, mod)
    return mod


def load_file_as(name: str, path: str, max_iter: int = 200):
    """exec a reference source file as module `name`, stubbing missing imports on the fly."""
    log = []
    for _ in range(max_iter):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        try:
            spec.loader.exec_module(mod)
            return mod, log
        except ModuleNotFoundError as e:
            sys.modules.pop(name, None)
            missing = e.name
            if missing == name:
                raise
            log.append(f"stub module {missing}")
            _stub_module(missing)
        except ImportError as e:
            sys.modules.pop(name, None)
            m = re.search(r"cannot import name '(\w+)' from '([\w\.]+)'", str(e))
            if not m:
                raise
            attr, modname = m.group(1), m.group(2)
            log.append(f"stub {modname}.{attr}")
            if modname not in sys.modules:
                importlib.import_module(modname)
            setattr(sys.modules[modname], attr, type(attr, (object,), {"__stub__": True}))
    raise RuntimeError("too many missing names: " + "; ".join(log[-10:]))


def load_ref_beam_search():
    """The reference's vendored BeamSearchScorer (indextts/gpt/transformers_beam_search.py)."""
    return load_file_as("ref_transformers_beam_search", f"{REF}/indextts/gpt/transformers_beam_search.py")


def _functional_presets():
    """Names that ARE exercised by generate(): give them their 4.52 behaviour instead of an inert stub."""
    import torch
    import transformers.pytorch_utils as pu
    if not hasattr(pu, "isin_mps_friendly"):
        pu.isin_mps_friendly = lambda elements, test_elements: torch.isin(elements, test_elements)
    import transformers.generation.configuration_utils as cu
    if not hasattr(cu, "NEED_SETUP_CACHE_CLASSES_MAPPING"):
        cu.NEED_SETUP_CACHE_CLASSES_MAPPING = {}
    if not hasattr(cu, "QUANT_BACKEND_CLASSES_MAPPING"):
        cu.QUANT_BACKEND_CLASSES_MAPPING = {}


def load_ref_generation_utils():
    """The reference's vendored GenerationMixin (indextts/gpt/transformers_generation_utils.py).

    It imports `transformers.generation.beam_search`, which 5.15 no longer ships: alias the reference's
    own vendored copy of that module.
    """
    _functional_presets()
    bs, log1 = load_ref_beam_search()
    sys.modules["transformers.generation.beam_search"] = bs
    import transformers.generation as tg
    tg.beam_search = bs
    mod, log2 = load_file_as("ref_transformers_generation_utils",
                             f"{REF}/indextts/gpt/transformers_generation_utils.py")
    return mod, log1 + log2


if __name__ == "__main__":
    m, log = load_ref_generation_utils()
    print("\n".join(log))
    print("loaded:", m.GenerationMixin)
