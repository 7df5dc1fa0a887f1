"""world_size-2 gloo test of the utterance-parallel path (runs on CPU): sharding covers every utterance exactly once,
the speaker bundle arrives bit-identical on every rank, waveforms are gathered in utterance order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from indextts_amd import dist as D
        n = 11
        lengths = [5, 90, 17, 64, 64, 3, 120, 8, 33, 70, 2]
        mine = D.shard_utterances(n, lengths=lengths)
        plain = D.shard_utterances(n)
        g = torch.Generator().manual_seed(0)
        bundle = None
        if rank == 0:
            bundle = {"style": torch.randn(1, 192, generator=g), "emo_vec": torch.randn(1, 1280, generator=g),
                      "spk_cond_emb": torch.randn(1, 37, 1024, generator=g), "ref_mel": torch.randn(1, 80, 50, generator=g)}
        got = D.broadcast_speaker_bundle(bundle, src=0)
        checksum = {k: float(v.double().sum()) for k, v in got.items()}
        wavs = [torch.full((10 + i,), i, dtype=torch.int16) for i in mine]
        all_w = D.gather_waveforms(wavs, mine, n, dst=0)
        ok_gather = None
        if rank == 0:
            ok_gather = all(all_w[i] is not None and all_w[i].numel() == 10 + i and int(all_w[i][0]) == i for i in range(n))
        # tensor gather with the deterministic LPT shard table (uneven shares: 6 and 5 rows) -- nothing but the blocks crosses the wire
        shards = [D.shard_utterances(n, r, world, lengths=lengths) for r in range(world)]
        block = torch.stack([torch.full((7,), 100 + i, dtype=torch.int16) for i in mine])
        full = D.gather_waveform_tensor(block, mine, n, dst=0, shards=shards)
        if rank == 0:
            ok_gather = ok_gather and full.shape == (n, 7) and [int(v) for v in full[:, 0]] == [100 + i for i in range(n)]
        else:
            assert full is None
        # a bundle larger than the flat buffer: every rank raises (no rank is left waiting in a collective)
        try:
            D.broadcast_speaker_bundle({"style": torch.zeros(1, 4096)} if rank == 0 else None, src=0, capacity=8192)
            overflow = False
        except ValueError:
            overflow = True
        assert overflow
        q.put((rank, mine, plain, checksum, sorted(got.keys()), ok_gather,
               sum(lengths[i] for i in mine)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_broadcast_gather():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    (_, m0, p0, c0, k0, g0, l0), (_, m1, p1, c1, k1, g1, l1) = res
    assert sorted(m0 + m1) == list(range(11)) and not set(m0) & set(m1)
    assert sorted(p0 + p1) == list(range(11)) and p0 == list(range(0, 11, 2))
    assert abs(l0 - l1) <= 4                       # LPT balance (rank::world would be 241 vs 235 here by luck; LPT is bounded)
    assert k0 == k1 == ["emo_vec", "ref_mel", "spk_cond_emb", "style"]
    assert c0 == c1
    assert g0 is True


class _RowGPT:
    """fake decoder whose codes depend on the row's own text only (so sharding must not change any row)"""
    n_text_pos = 42

    def __init__(self):
        self.calls = []

    def inference_speech(self, cond, text, langs, **kw):
        self.calls.append((text.clone(), kw))
        B = text.shape[0]
        codes = torch.full((B, 9), 8193, dtype=torch.long)
        for b in range(B):
            n = 3 + int((text[b] != 1).sum()) % 5
            codes[b, :n] = int(text[b][text[b] != 1].sum()) % 50 + torch.arange(n)      # padding-independent
        return codes, None


class _RowVoc:
    total_up = 256

    def __call__(self, mel, lens=None):
        B, _, T = mel.shape
        w = torch.zeros(B, 1, T * 256)
        for b in range(B):
            n = int(lens[b])
            w[b, :, : n * 256] = 0.5 * torch.tanh(mel[b, :, :n].mean())
        return w


def _make_rowwise():
    from indextts_amd.infer_v2_5 import IndexTTS2
    from tests.pipeline_stubs import StubFrontend
    fe = StubFrontend(64)
    return IndexTTS2(cfg={"gpt": {"stop_mel_token": 8193}}, device="cpu", frontend=fe, gpt=_RowGPT(), bigvgan=_RowVoc()), fe


def _pipeline_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tts, fe = _make_rowwise()
        res = tts.infer_batch("spk.wav", ["a. bb", "ccc", "d. ee. fff", "gggg"], "en", num_beams=1)
        speaker_calls = len([c for c in fe.calls if c[0] == "speaker"])
        rows = tts.gpt.calls[-1][0].shape[0] if tts.gpt.calls else 0
        q.put((rank, [None if r is None else (r[0], r[1].tobytes()) for r in res], speaker_calls, rows))
    finally:
        dist.destroy_process_group()


def test_infer_batch_shards_over_two_ranks():
    """IndexTTS2.infer_batch under a 2-rank group: rank 0 alone runs the prompt encoders, the 7 segment rows are split over
    the ranks, and rank 0 returns exactly what a single process returns."""
    tts, _ = _make_rowwise()
    want = tts.infer_batch("spk.wav", ["a. bb", "ccc", "d. ee. fff", "gggg"], "en", num_beams=1)
    want = [(r[0], r[1].tobytes()) for r in want]
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, out0, spk0, rows0), (_, out1, spk1, rows1) = res
    assert out0 == want
    assert out1 == [None] * 4
    assert spk0 == 1 and spk1 == 0                      # prompt encoders ran on rank 0 only
    assert rows0 + rows1 == 7 and rows0 > 0 and rows1 > 0
