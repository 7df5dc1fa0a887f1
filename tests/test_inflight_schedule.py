"""The in-flight batching schedule of `UnifiedVoice.inference_speech_inflight` (design reference: backends/trt/serving/triton_server.py:96-305,
backends/trt/pipeline/pipeline.py:459-548) on a SIMULATED decode session: every utterance's ids are a function of the utterance alone, so the test
can check that each one comes back complete, in order and exactly once whatever the slot count, poll interval and admission window -- and that the
schedule really reuses freed slots.  (The engine-level contract -- an admitted row's ids equal the row decoded alone -- is the GPU test
tests/test_gpu_admission.py; the schedule against the real engine: test_inflight_equals_one_batch there.)"""
import pytest
import torch

from indextts_amd import gpt, serving

STOP = 99


def _ids(utt, length):
    return [(7 * utt + 3 * t) % 90 for t in range(length)]


class _FakeSession:
    """Rows emit _ids(utt, len) then STOP for ever; one session step counter that runs for as long as it is asked to; every slot keeps its own
    step (`step0`) and writes its codes from column 0 of its code row (as the engine does); a row stores nothing past its max_new columns."""
    log = []

    def __init__(self, model, emb, mask, max_new, row_max_new=None, **kw):
        self.utts = [int(v) for v in emb[:, 0, 0].tolist()]
        self.lens = {u: int(emb[i, 0, 1]) for i, u in enumerate(self.utts)}
        if row_max_new is not None:                          # a capped row emits the stop token from its cap on, as the engine's sampler does
            for u, c in zip(self.utts, row_max_new):
                self.lens[u] = min(self.lens[u], int(c))
        self.capped = row_max_new is not None
        self.B, self.max_new, self.steps = len(self.utts), int(max_new), 0
        self.step0 = [0] * self.B
        self._codes = torch.full((self.B, self.max_new), STOP, dtype=torch.int64)
        self.S = emb.shape[1] + 1
        _FakeSession.log.append(("open", list(self.utts), self.max_new))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        _FakeSession.log.append(("close", self.steps))

    def run(self, n, return_when_finished=0):
        limit = self.steps + n if self.steps else min(self.max_new, n)
        for step in range(self.steps, limit):
            for b, u in enumerate(self.utts):
                t = step - self.step0[b]
                if t < self.max_new:
                    self._codes[b, t] = _ids(u, self.lens[u])[t] if t < self.lens[u] else STOP
            self.steps = step + 1
            if self.steps % 8 == 0 and self.steps < limit:       # the engine's flag check: every row finished, or enough slots to refill
                fin = len(self.finished())
                if fin == self.B or (return_when_finished > 0 and fin >= return_when_finished):
                    break
        return self.steps

    def _own(self, b):
        return self.steps - self.step0[b]

    def finished(self):
        return [b for b in range(self.B) if bool((self._codes[b, :min(self._own(b), self.max_new)] == STOP).any()) or self._own(b) > self.max_new]

    def codes(self, b):
        row = self._codes[b, :min(self._own(b), self.max_new)]
        hit = (row == STOP).nonzero()
        return row[: int(hit[0])].clone() if hit.numel() else row.clone()

    def finished_lengths(self):
        return [(b, int(self.codes(b).numel())) for b in self.finished()]

    def admit(self, slots, emb, mask, row_max_new=None):
        assert self.steps >= 1 and emb.shape[1] + 1 <= self.S
        assert (row_max_new is not None) == self.capped
        for b, i in zip(slots, range(emb.shape[0])):
            u = int(emb[i, 0, 0])
            assert b in self.finished(), "admitted into a slot whose row is still running"
            self.utts[b], self.lens[u], self.step0[b] = u, int(emb[i, 0, 1]), self.steps - 1
            if row_max_new is not None:
                self.lens[u] = min(self.lens[u], int(row_max_new[i]))
            self._codes[b, :] = STOP
            # the admitted row's first id is sampled by the admission prefill into column 0 of its code row
            self._codes[b, 0] = _ids(u, self.lens[u])[0] if self.lens[u] > 0 else STOP
        _FakeSession.log.append(("admit", list(slots), self.steps))


def _model(lengths, table=400):
    m = object.__new__(gpt.UnifiedVoice)
    m.kv_cache, m.stop_mel_token, m.device = True, STOP, "cpu"
    m._emb = {"mel_pos_embedding.emb.weight": torch.zeros(table + 1, 4)}          # -> `table` steps per session with kv_cache

    def prep(speech_condition, text_inputs, langs, cond_lengths, emo_vec, campplus_embedding, input_tokens, nret, max_generate_length, *rest):
        n = len(lengths)
        emb = torch.zeros(n, 5, 4)
        emb[:, 0, 0] = torch.arange(n, dtype=torch.float32)
        emb[:, 0, 1] = torch.tensor(lengths, dtype=torch.float32)
        return emb, torch.ones(n, 6, dtype=torch.long), int(max_generate_length), dict(rest[-1]), None
    m._prepare_inference = prep
    return m


@pytest.mark.parametrize("slots,chunk", [(1, 4), (2, 8), (3, 5), (4, 16), (8, 3)])
def test_every_utterance_comes_back_complete_and_in_order(monkeypatch, slots, chunk):
    monkeypatch.setattr(gpt, "DecodeSession", _FakeSession)
    lengths = [30, 3, 0, 17, 40, 8, 8, 1, 25, 12, 5]
    m = _model(lengths)
    _FakeSession.log = []
    codes, _ = m.inference_speech_inflight(None, None, max_generate_length=64, slots=slots, chunk_tokens=chunk, do_sample=False)
    assert codes.shape[0] == len(lengths) and codes.shape[1] == max(lengths) + 1
    for u, n in enumerate(lengths):
        assert codes[u, :n].tolist() == _ids(u, n) and bool((codes[u, n:] == STOP).all()), u
    st = m.last_inflight
    opened = [e for e in _FakeSession.log if e[0] == "open"]
    assert st["sessions"] == len(opened) and st["truncated"] == 0
    assert st["admitted"] + sum(len(e[1]) for e in opened) == len(lengths)          # every utterance entered exactly once
    if slots < len(lengths):
        assert st["admitted"] > 0                                                    # freed slots were reused
        assert all(len(e[1]) <= slots for e in opened)
    # in-flight beats draining: fewer decode steps than running ceil(N / slots) batches to their longest row
    drained = sum(max(lengths[i:i + slots]) + 1 for i in range(0, len(lengths), slots))
    assert st["steps"] <= drained + chunk * len(opened) + chunk * st["admitted"]


def test_row_budget_is_per_row_and_one_session_serves_the_call(monkeypatch):
    """The mel position table bounds a ROW, not the session: with a table of 60 steps and 30-token budgets, six utterances on two slots run in ONE
    session whose step counter passes the table, every admitted utterance gets its full budget, and the default call (no `row_max_new`, no
    scheduling arguments) does admit."""
    monkeypatch.setattr(gpt, "DecodeSession", _FakeSession)
    lengths = [50, 20, 20, 20, 20, 20]
    m = _model(lengths, table=60)
    _FakeSession.log = []
    codes, _ = m.inference_speech_inflight(None, None, max_generate_length=30, slots=2, chunk_tokens=4, do_sample=False)
    st = m.last_inflight
    assert st["truncated"] == 1                              # utterance 0 runs into its own 30-token budget: no stop token of its own
    assert codes.shape[1] == 30 and codes[0].tolist() == _ids(0, 50)[:30]
    for u in range(1, 6):
        assert codes[u, :20].tolist() == _ids(u, 20) and int(codes[u, 20]) == STOP
    assert st["sessions"] == 1 and st["admitted"] == 4 and len([e for e in _FakeSession.log if e[0] == "open"]) == 1
    assert st["steps"] > 60                                  # past the table: 5 x 20-token rows + polls on one of the slots
    assert max(e[2] for e in _FakeSession.log if e[0] == "admit") + 30 > 60          # an utterance joined where the old shared counter had no room left
    with pytest.raises(ValueError):
        m.inference_speech_inflight(None, None, max_generate_length=61, slots=2)
    with pytest.raises(NotImplementedError):
        m.inference_speech_inflight(None, None, max_generate_length=30, slots=2, num_beams=3)


def test_default_serving_budget_admits(monkeypatch):
    """The pipeline's defaults (max_mel_tokens 1500 under a 1815-row table; `infer_batch(inflight_slots=)` passes no caps): admission must fire for
    every waiting utterance -- the budget of a late one is its own."""
    monkeypatch.setattr(gpt, "DecodeSession", _FakeSession)
    lengths = [700, 300, 650, 400, 500, 620, 80, 900, 800, 750, 640]
    m = _model(lengths, table=1815)
    _FakeSession.log = []
    codes, _ = m.inference_speech_inflight(None, None, max_generate_length=1500, slots=3, chunk_tokens=16, do_sample=False)
    st = m.last_inflight
    assert st["sessions"] == 1 and st["admitted"] == 8 and st["truncated"] == 0
    for u, n in enumerate(lengths):
        assert codes[u, :n].tolist() == _ids(u, n) and int(codes[u, n]) == STOP
    assert st["steps"] > 1815                                # the session outlives the table


def test_batcher_passes_the_slot_count_for_single_beam_requests():
    class TTS:
        calls = []

        def infer_batch(self, spk, texts, lang, emo_audio_prompt=None, emo_alpha=1.0, **gen):
            TTS.calls.append(dict(gen))
            return [(22050, None)] * len(texts)
    b = serving.DynamicBatcher(TTS(), max_batch=4, max_wait_ms=5, inflight_slots=2)
    b.submit(b"A", "one", "en", num_beams=1).result(timeout=10)
    b.submit(b"A", "two", "en").result(timeout=10)                                   # reference default: 3 beams -> the plain batch path
    b.close()
    assert TTS.calls == [{"num_beams": 1, "inflight_slots": 2}, {}]


def test_per_utterance_caps_and_admission_batching(monkeypatch):
    monkeypatch.setattr(gpt, "DecodeSession", _FakeSession)
    lengths = [40] * 12                                      # nothing stops on its own within the caps
    caps = [5, 33, 9, 12, 30, 7, 21, 3, 16, 11, 8, 2]
    m = _model(lengths)
    _FakeSession.log = []
    codes, _ = m.inference_speech_inflight(None, None, max_generate_length=36, slots=4, chunk_tokens=4, min_free=2, row_max_new=caps, do_sample=False)
    for u, n in enumerate(caps):
        assert codes[u, :n].tolist() == _ids(u, 40)[:n] and bool((codes[u, n:] == STOP).all()), u
    st = m.last_inflight
    assert st["truncated"] == 12 and st["admitted"] == 8 and st["admissions"] <= 4         # every row ends at its cap; at least two slots per admission
    for e in _FakeSession.log:
        if e[0] == "admit":
            assert len(e[1]) >= 2 or e is [x for x in _FakeSession.log if x[0] == "admit"][-1]
    with pytest.raises(ValueError):
        m.inference_speech_inflight(None, None, max_generate_length=36, slots=4, row_max_new=caps[:3])


def test_session_finished_and_codes_respect_the_slot_steps():
    """`DecodeSession.finished()` / `codes()` look only at the columns the utterance that occupies the slot NOW has produced (its own steps, from
    `step0` on): what sits further right in the row does not count; a row past its last column counts as finished."""
    class M:
        stop_mel_token = STOP
    s = object.__new__(gpt.DecodeSession)
    s.m, s.dev, s.B, s.max_new = M(), "cpu", 4, 8
    s._codes = torch.tensor([[1, 2, STOP, STOP, STOP, STOP, STOP, STOP],      # slot 0: first occupant stopped after 2 ids
                             [1, 2, 3, 4, 5, 6, 7, 8],                          # slot 1: still running
                             [7, 8, 9, STOP, STOP, STOP, STOP, STOP],           # slot 2: an utterance admitted at step 3 (its first id), stopped after 3 ids
                             [5, 6, 7, 8, STOP, STOP, STOP, STOP]], dtype=torch.int64)   # slot 3: admitted at step 4, still running
    s.step0 = [0, 0, 3, 4]
    s.steps = 8
    assert s.finished() == [0, 2]
    assert s.codes(0).tolist() == [1, 2] and s.codes(2).tolist() == [7, 8, 9] and s.codes(3).tolist() == [5, 6, 7, 8] and s.codes(1).numel() == 8
    s.steps = 5                                                                  # earlier in time: slot 2 has produced 7, 8 so far, slot 3 its first id
    assert s.finished() == [0] and s.codes(2).tolist() == [7, 8] and s.codes(3).tolist() == [5]
    assert s.finished_lengths() == [(0, 2)]
    s.steps = 9                                                                  # slot 1 is past its last column: the engine has it as stopped
    assert s.finished() == [0, 1, 2, 3] and s.codes(1).numel() == 8 and s.codes(3).tolist() == [5, 6, 7, 8]      # (slot 3's fifth column holds a stop token)
    assert s.finished_lengths() == [(0, 2), (1, 8), (2, 3), (3, 4)]
    s.steps = 0
    assert s.finished() == []
