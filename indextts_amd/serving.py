"""In-process serving shell: a speaker-bundle cache keyed by the prompt audio's bytes and a dynamic
batcher that turns concurrent single-utterance requests into `IndexTTS2.infer_batch` calls -- the pieces of the reference's
Triton front end that decide WHAT runs as one batch (`SpeakerCache`, backends/trt/serving/triton_server.py:44-94; the `@batch`
decorated `infer_non_streaming`, :170-230), without its network layer.  The engine batches utterances of ONE speaker bundle, so
requests are grouped by (speaker prompt, emotion prompt, emo_alpha, language, generation settings)."""
import collections
import hashlib
import threading
import time
from concurrent.futures import Future
from typing import Any, Callable, Dict, Hashable, List, Optional, Tuple


class SpeakerCache:
    """LRU of computed speaker bundles keyed by a hash of the prompt audio (bytes or a path): the prompt encoders run once per
    distinct prompt (triton_server.py:44-94)."""

    def __init__(self, compute: Callable[[Any], Any], max_size: int = 64):
        self.compute, self.max_size = compute, int(max_size)
        self._d: "collections.OrderedDict[str, Any]" = collections.OrderedDict()
        self._lock = threading.Lock()
        self.hits = self.misses = 0

    @staticmethod
    def key_of(audio) -> str:
        data = audio if isinstance(audio, (bytes, bytearray, memoryview)) else str(audio).encode()
        return hashlib.sha256(bytes(data)).hexdigest()

    def get_or_compute(self, audio):
        k = self.key_of(audio)
        with self._lock:
            if k in self._d:
                self._d.move_to_end(k)
                self.hits += 1
                return self._d[k]
        v = self.compute(audio)                      # outside the lock: prompt encoders take a while
        with self._lock:
            self.misses += 1
            self._d[k] = v
            self._d.move_to_end(k)
            while len(self._d) > self.max_size:
                self._d.popitem(last=False)
        return v


class _Request:
    __slots__ = ("group", "text", "future", "t")

    def __init__(self, group, text):
        self.group, self.text, self.future, self.t = group, text, Future(), time.monotonic()


class DynamicBatcher:
    """submit(...) returns a Future of `(sample_rate, int16 array)`; a worker thread collects the requests of one group until
    `max_batch` of them wait or the oldest has waited `max_wait_ms`, and runs them as ONE `infer_batch` call.  Groups are served
    oldest-first; a failing batch fails exactly its own futures."""

    def __init__(self, tts, max_batch: int = 64, max_wait_ms: float = 10.0, inflight_slots: Optional[int] = None):
        """inflight_slots: decode at most that many rows at a time and admit the batch's waiting utterances into the slots of rows that have
        stopped (`UnifiedVoice.inference_speech_inflight`, num_beams = 1) -- `max_batch` can then exceed what one decode batch should hold."""
        self.tts, self.max_batch, self.max_wait = tts, int(max_batch), float(max_wait_ms) / 1000.0
        self.inflight_slots = None if not inflight_slots else int(inflight_slots)
        self._q: List[_Request] = []
        self._cv = threading.Condition()
        self._stop = False
        self.batches: List[int] = []                 # size of every batch that ran (observability / tests)
        self._worker = threading.Thread(target=self._run, name="indextts-batcher", daemon=True)
        self._worker.start()

    def submit(self, spk_audio_prompt, text: str, lang, emo_audio_prompt=None, emo_alpha: float = 1.0, **generation_kwargs) -> Future:
        group: Tuple[Hashable, ...] = (SpeakerCache.key_of(spk_audio_prompt), None if emo_audio_prompt is None else SpeakerCache.key_of(emo_audio_prompt),
                                       float(emo_alpha), lang, tuple(sorted(generation_kwargs.items())))
        r = _Request((group, spk_audio_prompt, emo_audio_prompt), text)
        with self._cv:
            if self._stop:
                raise RuntimeError("DynamicBatcher is closed")
            self._q.append(r)
            self._cv.notify()
        return r.future

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify()
        self._worker.join()

    def _take(self) -> Optional[List[_Request]]:
        with self._cv:
            while True:
                if self._q:
                    head = self._q[0]
                    same = [r for r in self._q if r.group[0] == head.group[0]]
                    wait = self.max_wait - (time.monotonic() - head.t)
                    if len(same) >= self.max_batch or wait <= 0 or self._stop:
                        take = same[: self.max_batch]
                        ids = {id(r) for r in take}
                        self._q = [r for r in self._q if id(r) not in ids]
                        return take
                    self._cv.wait(timeout=wait)
                elif self._stop:
                    return None
                else:
                    self._cv.wait()

    def _run(self):
        while True:
            reqs = self._take()
            if reqs is None:
                return
            (key, spk, emo) = reqs[0].group
            _, _, emo_alpha, lang, gen = key
            self.batches.append(len(reqs))
            try:
                gen = dict(gen)
                if self.inflight_slots and gen.get("num_beams", 3) == 1:
                    gen.setdefault("inflight_slots", self.inflight_slots)
                res = list(self.tts.infer_batch(spk, [r.text for r in reqs], lang, emo_audio_prompt=emo, emo_alpha=emo_alpha, **gen))
                if len(res) != len(reqs):
                    raise RuntimeError(f"infer_batch returned {len(res)} results for {len(reqs)} requests")
                for r, out in zip(reqs, res):
                    if r.future.set_running_or_notify_cancel():      # a caller may have cancelled while the batch ran
                        r.future.set_result(out)
            except Exception as e:                    # noqa: BLE001 -- delivered to the callers of this batch
                for r in reqs:
                    if not r.future.done():
                        try:
                            r.future.set_exception(e)
                        except Exception:             # noqa: BLE001 -- cancelled in between: nobody is waiting
                            pass


def synthesize_tasks(tts, tasks: List[Dict[str, Any]], lang=None, max_batch: int = 64, **generation_kwargs) -> List[str]:
    """Batch-file synthesis (`_run_batch`, indextts/cli_v2.py:605-678, which calls `tts.infer` once per task): tasks that share
    the voice prompt and emotion settings run as real `infer_batch` batches of up to `max_batch` utterances; every task's audio
    is written to its own `output_path` (16-bit PCM WAV, `save_pcm_wav` semantics).  A task is a dict with `voice_path`, `text`,
    `output_path` and optional `emotion_kwargs` (`emo_audio_prompt`, `emo_alpha`) / `line_number`, as `_load_batch_tasks` builds
    them.  Returns the written paths in task order; a failing batch raises with the line numbers it covered."""
    import os
    import torch
    from .infer_v2_5 import save_pcm_wav
    groups: "collections.OrderedDict[Tuple, List[int]]" = collections.OrderedDict()
    for i, t in enumerate(tasks):
        ek = dict(t.get("emotion_kwargs") or {})
        unsupported = set(ek) - {"emo_audio_prompt", "emo_alpha"}
        if unsupported:
            raise ValueError(f"batch line {t.get('line_number', i + 1)}: {sorted(unsupported)} need the per-utterance infer() path")
        key = (str(t["voice_path"]), None if ek.get("emo_audio_prompt") is None else str(ek["emo_audio_prompt"]), float(ek.get("emo_alpha", 1.0)))
        groups.setdefault(key, []).append(i)
    written: List[Optional[str]] = [None] * len(tasks)
    for (voice, emo, alpha), idx in groups.items():
        for j in range(0, len(idx), max_batch):
            part = idx[j: j + max_batch]
            try:
                res = tts.infer_batch(voice, [tasks[i]["text"] for i in part], lang, emo_audio_prompt=emo, emo_alpha=alpha, **generation_kwargs)
            except Exception as e:                # noqa: BLE001
                lines = [tasks[i].get("line_number", i + 1) for i in part]
                raise RuntimeError(f"batch file lines {lines} inference failed: {e}") from e
            for i, out in zip(part, res):
                if out is None:
                    continue
                path = str(tasks[i]["output_path"])
                os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
                sr, wav = out
                save_pcm_wav(path, torch.from_numpy(wav.T.copy()).float(), sr)
                written[i] = path
    return written
