"""Streaming shell: `indextts_amd.streaming.StreamingDecoder` against the reference's own
`StreamingDecoder.generate` (tests/golden/streaming.npz, minted by tools/make_golden_streaming.py from
backends/trt/pipeline/streaming.py): same chunk script, same rendered audio -> bit-identical int16 pieces, done flags and number
of yields; plus the dynamic batcher's grouping rules."""
import os

import numpy as np
import pytest

from indextts_amd import streaming
from tools.make_golden_streaming import CASES


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_streaming_decoder_matches_reference(golden_dir, ci):
    z = np.load(os.path.join(golden_dir, "streaming.npz"))
    case = CASES[ci]
    B = case["B"]
    audio_in = [[z[f"c{ci}_in{k}_a{b}"] for b in range(B)] for k in range(len(case["chunks"]))]

    class Engine:
        def generate_chunks(self, inputs_embeds, attention_mask, max_new_tokens, chunk_size, overlap_size, **kw):
            assert (chunk_size, overlap_size) == (case["chunk"], case["overlap"])
            for n, last, done, lens in case["chunks"]:
                yield np.zeros((B, n), dtype=np.int64), last, done, np.asarray(lens)

    it = iter(audio_in)
    dec = streaming.StreamingDecoder(Engine(), lambda codes, lens: next(it), chunk_size=case["chunk"], overlap_size=case["overlap"])
    res = list(dec.generate(np.zeros((B, 1, 1)), None, 0))
    assert len(res) == int(z[f"c{ci}_n_yield"])
    for yi, (sr, audio, done) in enumerate(res):
        assert sr == 22050 and list(done) == z[f"c{ci}_y{yi}_done"].tolist()
        for b in range(B):
            assert (audio[b] is not None) == bool(z[f"c{ci}_y{yi}_has{b}"])
            if audio[b] is not None:
                ref = z[f"c{ci}_y{yi}_a{b}"]
                assert audio[b].dtype == np.int16 and np.array_equal(audio[b], ref), (ci, yi, b)
    assert dec.first_chunk_latency is not None


def test_overlap_must_be_smaller_than_chunk():
    with pytest.raises(ValueError):
        streaming.StreamingDecoder(object(), lambda c, n: [], chunk_size=10, overlap_size=10)
    assert streaming.overlap_samples(20) == int(20 * 1.72) * 256


class _FakeTTS:
    def __init__(self, delay=0.0, fail_on=None):
        self.calls, self.delay, self.fail_on = [], delay, fail_on

    def infer_batch(self, spk, texts, lang, emo_audio_prompt=None, emo_alpha=1.0, **gen):
        import time
        self.calls.append((spk, list(texts), lang, emo_audio_prompt, emo_alpha, dict(gen)))
        time.sleep(self.delay)
        if self.fail_on is not None and self.fail_on in texts:
            raise RuntimeError("boom")
        return [(22050, np.full((len(t), 1), len(spk), dtype=np.int16)) for t in texts]


def test_dynamic_batcher_groups_by_speaker_and_settings():
    from indextts_amd.serving import DynamicBatcher
    tts = _FakeTTS(delay=0.05)
    b = DynamicBatcher(tts, max_batch=4, max_wait_ms=200)
    futs = [b.submit(b"speaker-A", f"text {i}", "en") for i in range(6)]            # 4 + 2 (max_batch)
    futs += [b.submit(b"speaker-BB", "other voice", "en")]                          # another group
    futs += [b.submit(b"speaker-A", "hot", "en", temperature=1.2)]                  # same voice, other settings: its own batch
    outs = [f.result(timeout=10) for f in futs]
    b.close()
    assert [o[1].shape[0] for o in outs] == [len(f"text {i}") for i in range(6)] + [len("other voice"), len("hot")]
    assert int(outs[6][1][0, 0]) == len(b"speaker-BB") and int(outs[0][1][0, 0]) == len(b"speaker-A")
    sizes = sorted(len(c[1]) for c in tts.calls)
    assert sizes == [1, 1, 2, 4] and b.batches == [len(c[1]) for c in tts.calls]
    assert [c[5] for c in tts.calls if c[1] == ["hot"]] == [{"temperature": 1.2}]
    for c in tts.calls:                                                              # a batch never mixes groups
        assert len({t.startswith("text") for t in c[1]}) == 1


def test_dynamic_batcher_failure_is_local_to_its_batch():
    from indextts_amd.serving import DynamicBatcher
    tts = _FakeTTS(fail_on="bad")
    b = DynamicBatcher(tts, max_batch=8, max_wait_ms=30)
    bad = b.submit(b"A", "bad", "en")
    with pytest.raises(RuntimeError, match="boom"):
        bad.result(timeout=10)
    ok = b.submit(b"A", "fine", "en")
    assert ok.result(timeout=10)[0] == 22050
    b.close()
    with pytest.raises(RuntimeError):
        b.submit(b"A", "late", "en")


def test_speaker_cache_lru():
    from indextts_amd.serving import SpeakerCache
    seen = []
    c = SpeakerCache(lambda a: (seen.append(a), len(a))[1], max_size=2)
    assert [c.get_or_compute(x) for x in (b"a", b"bb", b"a", b"ccc", b"bb")] == [1, 2, 1, 3, 2]
    assert seen == [b"a", b"bb", b"ccc", b"bb"] and (c.hits, c.misses) == (1, 4)


def test_synthesize_tasks_runs_real_batches(tmp_path):
    import wave
    from indextts_amd.serving import synthesize_tasks
    tts = _FakeTTS()
    tasks = [dict(voice_path="a.wav", text=f"line {i}", output_path=tmp_path / "out" / f"{i}.wav", line_number=i + 1) for i in range(5)]
    tasks.insert(2, dict(voice_path="b.wav", text="other", output_path=tmp_path / "o.wav", emotion_kwargs={"emo_alpha": 0.5}, line_number=9))
    paths = synthesize_tasks(tts, tasks, lang="en", max_batch=4, top_k=5)
    assert [len(c[1]) for c in tts.calls] == [4, 1, 1]                      # a.wav: 4 + 1, b.wav: 1 -- not six sequential calls
    assert tts.calls[2][0] == "b.wav" and tts.calls[2][4] == 0.5 and tts.calls[0][5] == {"top_k": 5}
    assert paths == [str(t["output_path"]) for t in tasks]
    with wave.open(paths[0], "rb") as w:
        assert (w.getframerate(), w.getsampwidth(), w.getnchannels(), w.getnframes()) == (22050, 2, 1, len("line 0"))
    with pytest.raises(ValueError, match="per-utterance"):
        synthesize_tasks(tts, [dict(voice_path="a", text="t", output_path=tmp_path / "x.wav", emotion_kwargs={"emo_text": "sad"})])
