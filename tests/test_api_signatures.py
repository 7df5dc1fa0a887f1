"""The kept Python API: constructor / infer / infer_generator signatures of the three pipeline classes equal the
reference's, parameter by parameter and default by default (fixture minted from the reference SOURCE by tools/make_golden_api.py)."""
import inspect
import json
import os

import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "api_signatures.json")))


def _cls(key):
    fname = key.split(":")[0]
    if fname == "infer.py":
        from indextts_amd.infer import IndexTTS as C
    elif fname == "infer_v2.py":
        from indextts_amd.infer_v2 import IndexTTS2 as C
    else:
        from indextts_amd.infer_v2_5 import IndexTTS2 as C
    return C


@pytest.mark.parametrize("key", sorted(GOLD))
def test_signature_matches_reference(key):
    want = GOLD[key]
    fn = getattr(_cls(key), key.split(".")[-1])
    ps = list(inspect.signature(fn).parameters.values())[1:]                       # drop self
    pos = [p for p in ps if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    assert [p.name for p in pos] == [n for n, _ in want["params"]], key
    for p, (name, default) in zip(pos, want["params"]):
        if default is None:
            assert p.default is inspect.Parameter.empty, (key, name)
        else:
            assert p.default == eval(default), (key, name, p.default, default)      # literals only (None, numbers, strings, booleans)
    var_kw = [p.name for p in ps if p.kind == p.VAR_KEYWORD]
    assert (var_kw[0] if var_kw else None) == want["var_kw"], key
    # anything this engine adds is keyword-only (injection points for tests / integration), never positional
    assert all(p.kind == p.KEYWORD_ONLY for p in ps if p not in pos and p.kind != p.VAR_KEYWORD)


def test_v2_positional_call_binds_like_the_reference():
    """the 4th positional argument of the v2 `infer` is `emo_audio_prompt` (infer_v2.py:371-375), not v2.5's `lang`."""
    from indextts_amd.infer_v2 import IndexTTS2
    b = inspect.signature(IndexTTS2.infer).bind(None, "spk.wav", "text", None, "emo.wav", 0.6)
    assert b.arguments["emo_audio_prompt"] == "emo.wav" and b.arguments["emo_alpha"] == 0.6 and "lang" not in b.arguments


def test_low_vram_split_rule():
    """infer_v2_5.py:466-487 (pinned against the reference's own function when this file was written: tools/make_golden_api.py
    docstring; the expected pieces below are its outputs)."""
    from indextts_amd.infer_v2_5 import IndexTTS2
    f = IndexTTS2.split_text_by_punctuation
    assert f("", 40) == []
    assert f("no punctuation in this very long string of characters without any breaks at all", 40) == \
        ["no punctuation in this very long string of characters without any breaks at all"]
    assert f("Hello, world. This is a test! Really? yes; ok: fine", 20) == ["Hello, world.", " This is a test!", " Really? yes; ok:", " fine"]
