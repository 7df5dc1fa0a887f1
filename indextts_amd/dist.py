"""Utterance data parallelism for the hot path: one process per GPU, RCCL over xGMI (`backend="nccl"` on ROCm).

The reference has no distributed runtime at all ("run one server process per GPU").  Utterances (and the
text segments of one utterance) are independent given the speaker/emotion conditioning, so the path shards with no
data-path collective: each rank decodes and vocodes its own utterances; the ONLY collective is one broadcast of the
speaker bundle per batch from the rank that ran the prompt encoders, plus an optional gather of the int16 waveforms.
Nothing here runs inside the per-token loop.
"""
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_utterances(n_utts: int, rank_: Optional[int] = None, world_: Optional[int] = None,
                     lengths: Optional[Sequence[int]] = None) -> List[int]:
    """Indices of the utterances this rank owns.  With `lengths` (e.g. text token counts ~ decode steps) utterances are
    assigned longest-first to the least-loaded rank (LPT; deterministic, identical on every rank) so ranks finish
    together; otherwise `rank::world`."""
    r = rank() if rank_ is None else rank_
    w = world() if world_ is None else world_
    if lengths is None:
        return list(range(r, n_utts, w))
    order = sorted(range(n_utts), key=lambda i: (-int(lengths[i]), i))
    load = [0] * w
    owner = [0] * n_utts
    for i in order:
        tgt = min(range(w), key=lambda k: (load[k], k))
        owner[i] = tgt
        load[tgt] += int(lengths[i])
    return [i for i in range(n_utts) if owner[i] == r]


# fixed key order + shapes so every rank can allocate the receive buffers without a metadata exchange
BUNDLE_KEYS = ("style", "emo_vec", "spk_cond_emb", "emo_cond_emb", "ref_mel", "prompt_condition")


def broadcast_speaker_bundle(bundle: Optional[Dict[str, torch.Tensor]], src: int = 0, device=None) -> Dict[str, torch.Tensor]:
    """One broadcast per tensor of the per-speaker conditioning computed on `src` (SURVEY.md section 8e payload list).

    Shapes differ per reference clip, so `src` first broadcasts a small int64 header (ndim + dims per key, -1 for an
    absent key); the payload follows as float32.  <= ~9 MB per speaker; latency-bound on xGMI.
    """
    if world() == 1:
        assert bundle is not None
        return bundle
    r = rank()
    hdr = torch.full((len(BUNDLE_KEYS), 5), -1, dtype=torch.int64, device=device)
    if r == src:
        for i, k in enumerate(BUNDLE_KEYS):
            t = bundle.get(k) if bundle else None
            if t is not None:
                assert t.dim() <= 4
                hdr[i, 0] = t.dim()
                for d, s in enumerate(t.shape):
                    hdr[i, 1 + d] = s
    dist.broadcast(hdr, src)
    out: Dict[str, torch.Tensor] = {}
    for i, k in enumerate(BUNDLE_KEYS):
        nd = int(hdr[i, 0])
        if nd < 0:
            continue
        shape = [int(x) for x in hdr[i, 1:1 + nd]]
        if r == src:
            t = bundle[k].to(device=device, dtype=torch.float32).contiguous()
        else:
            t = torch.empty(shape, dtype=torch.float32, device=device)
        dist.broadcast(t, src)
        out[k] = t
    return out


def gather_waveforms(wavs: List[torch.Tensor], indices: List[int], n_utts: int, dst: int = 0):
    """Collect per-utterance int16 waveforms on `dst` in utterance order (44 KB per audio-second)."""
    if world() == 1:
        return [w for _, w in sorted(zip(indices, wavs))]
    payload = [(int(i), w.detach().to("cpu", torch.int16)) for i, w in zip(indices, wavs)]
    gathered = [None] * world() if rank() == dst else None
    dist.gather_object(payload, gathered, dst=dst)
    if rank() != dst:
        return None
    out = [None] * n_utts
    for part in gathered:
        for i, w in part:
            out[i] = w
    return out


def gather_waveform_tensor(wav: torch.Tensor, indices: Sequence[int], n_utts: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Tensor form of `gather_waveforms` for equal-length rows: every rank hands its (n_local, T) int16 block (device tensor
    for RCCL, CPU tensor for gloo) to `dst`, which returns the (n_utts, T) batch in utterance order.  One `gather` of
    44 KB per audio-second -- the only collective after the speaker-bundle broadcast, and like it outside the decode loop.
    Ranks may own different numbers of rows (blocks are padded to the largest shard; shard sizes follow from
    `shard_utterances`, so no size exchange is needed when `counts` is deterministic -- they are exchanged here once as a
    small int tensor to keep the function self-contained)."""
    w = world()
    if w == 1:
        out = torch.empty((n_utts,) + tuple(wav.shape[1:]), dtype=wav.dtype, device=wav.device)
        out[torch.as_tensor(list(indices), dtype=torch.long, device=wav.device)] = wav
        return out
    dev = wav.device
    r = rank()
    meta = torch.full((n_utts + 1,), -1, dtype=torch.int64, device=dev)
    meta[0] = len(indices)
    meta[1:1 + len(indices)] = torch.as_tensor(list(indices), dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(w)]
    dist.all_gather(metas, meta)
    cmax = max(int(m[0]) for m in metas)
    block = torch.zeros((cmax,) + tuple(wav.shape[1:]), dtype=wav.dtype, device=dev)
    block[: wav.shape[0]] = wav
    # bytes on the wire: neither RCCL nor gloo carries int16
    wire = block.view(torch.uint8)
    parts_b = [torch.empty_like(wire) for _ in range(w)] if r == dst else None
    dist.gather(wire, parts_b, dst=dst)
    if r != dst:
        return None
    parts = [p.view(wav.dtype) for p in parts_b]
    out = torch.empty((n_utts,) + tuple(wav.shape[1:]), dtype=wav.dtype, device=dev)
    for m, part in zip(metas, parts):
        c = int(m[0])
        if c:
            out[m[1:1 + c].to(torch.long)] = part[:c]
    return out
