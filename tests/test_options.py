"""The engine's option table (itts_set_option / itts_get_option): CPU-side contract -- no GPU, no compute."""
import re
import os

import pytest

from indextts_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_option_table_roundtrip_and_validation():
    opts = _lib.options()
    assert len(opts) >= 20 and "decode_fuse_ln" in opts and "attn_waves" in opts
    for name, (cur, default, doc) in opts.items():
        assert cur == default, name                    # nothing leaks between tests
        assert len(doc) > 10
    _lib.set_option("decode_fuse_ln", 0)
    assert _lib.get_option("decode_fuse_ln") == 0
    _lib.reset_options()
    assert _lib.get_option("decode_fuse_ln") == 1
    with pytest.raises(_lib.HipEngineError):
        _lib.set_option("no_such_option", 1)
    with pytest.raises(_lib.HipEngineError):
        _lib.set_option("x3_products", 99)
    # discrete sets: values inside the range that no launcher instantiates are refused (they used to be mapped differently by different launchers)
    for name, bad in (("x3_products", 7), ("decode_nt", 3), ("fa_qs", 3), ("conv_bm", 48), ("attn_waves", 5), ("decode_ln_nt", 3)):
        with pytest.raises(_lib.HipEngineError):
            _lib.set_option(name, bad)
    with _lib.option_scope(tile256=2, fa32_qs=1):
        assert _lib.get_option("tile256") == 2 and _lib.get_option("fa32_qs") == 1
    assert _lib.get_option("tile256") == -1 and _lib.get_option("fa32_qs") == 2


def test_header_documents_every_option_and_the_library_reads_no_environment():
    hdr = open(os.path.join(ROOT, "include", "indextts_hip.h")).read()
    for name, (_, default, _) in _lib.options().items():
        m = re.search(r"\*\s+" + re.escape(name) + r"\s+(-?\d+)\s", hdr)
        assert m, f"option {name} is not in the header's table"
        assert int(m.group(1)) == default, name
    import glob
    for src in glob.glob(os.path.join(ROOT, "indextts_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "indextts_amd", "csrc", "*.h")):
        assert "getenv" not in open(src).read(), f"{src} reads the environment"
