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


# fixed key order so that the header of the flat buffer is position-coded
BUNDLE_KEYS = ("style", "emo_vec", "spk_cond_emb", "emo_cond_emb", "ref_mel", "prompt_condition")
_HDR_INTS = len(BUNDLE_KEYS) * 5 + 1                                # (ndim, 4 dims) per key + the payload length in floats
_HDR_BYTES = (_HDR_INTS * 8 + 255) // 256 * 256
# room for the largest bundle the pipeline produces: the reference truncates reference audio to 15 s (infer_v2.py:466-470), i.e. 750-frame
# w2v-bert / emotion sequences (1024 wide), a 1292-frame mel (80) and prompt condition (512), a 192-wide style vector: 9.2 MB of float32
BUNDLE_CAPACITY = 12 << 20


def _group_active() -> bool:
    return dist.is_available() and dist.is_initialized()


def broadcast_speaker_bundle(bundle: Optional[Dict[str, torch.Tensor]], src: int = 0, device=None,
                             capacity: int = BUNDLE_CAPACITY) -> Dict[str, torch.Tensor]:
    """ONE broadcast of the per-speaker conditioning computed on `src` (SURVEY.md section 8e payload list): a flat byte buffer of fixed
    `capacity` = [int64 header: ndim + dims per key (-1 for an absent key), payload length | float32 payload of every present tensor in
    BUNDLE_KEYS order].  Shapes differ per reference clip, the capacity does not, so the receivers post their buffer without a size
    exchange; <= ~9 MB per speaker, outside the decode loop.  Runs whenever a process group exists (world size 1 included: the RCCL path
    is then exercised on one device); without a group it returns `bundle`."""
    if not _group_active():
        assert bundle is not None
        return bundle
    r = rank()
    buf = torch.empty(capacity, dtype=torch.uint8, device=device)
    hdr = torch.full((_HDR_INTS,), -1, dtype=torch.int64)
    if r == src:
        parts = []
        n_float = 0
        for i, k in enumerate(BUNDLE_KEYS):
            t = bundle.get(k) if bundle else None
            if t is None:
                continue
            assert t.dim() <= 4
            hdr[5 * i] = t.dim()
            for d, s_ in enumerate(t.shape):
                hdr[5 * i + 1 + d] = s_
            parts.append(t.detach().to(device=device, dtype=torch.float32).reshape(-1))
            n_float += t.numel()
        hdr[-1] = n_float
        if _HDR_BYTES + 4 * n_float > capacity:
            hdr[-1] = -2                                             # every rank raises instead of deadlocking on a short buffer
        else:
            buf[_HDR_BYTES:_HDR_BYTES + 4 * n_float].view(torch.float32).copy_(torch.cat(parts) if parts else torch.empty(0, device=device))
        buf[: _HDR_INTS * 8].view(torch.int64).copy_(hdr)
    dist.broadcast(buf, src)
    hdr = buf[: _HDR_INTS * 8].view(torch.int64).cpu()
    if int(hdr[-1]) == -2:
        raise ValueError(f"broadcast_speaker_bundle: the bundle does not fit the {capacity}-byte buffer (pass a larger `capacity` on every rank)")
    payload = buf[_HDR_BYTES:_HDR_BYTES + 4 * int(hdr[-1])].view(torch.float32)
    out: Dict[str, torch.Tensor] = {}
    at = 0
    for i, k in enumerate(BUNDLE_KEYS):
        nd = int(hdr[5 * i])
        if nd < 0:
            continue
        shape = [int(x) for x in hdr[5 * i + 1:5 * i + 1 + nd]]
        n = 1
        for s_ in shape:
            n *= s_
        out[k] = payload[at:at + n].view(shape)
        at += n
    return out


def gather_waveforms(wavs: List[torch.Tensor], indices: List[int], n_utts: int, dst: int = 0):
    """Collect per-utterance int16 waveforms on `dst` in utterance order (44 KB per audio-second)."""
    if not _group_active():
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


def gather_waveform_tensor(wav: torch.Tensor, indices: Sequence[int], n_utts: int, dst: int = 0,
                           shards: Optional[Sequence[Sequence[int]]] = None) -> Optional[torch.Tensor]:
    """Tensor form of `gather_waveforms` for equal-length rows: every rank hands its (n_local, T) int16 block (device tensor
    for RCCL, CPU tensor for gloo) to `dst`, which returns the (n_utts, T) batch in utterance order.  One `gather` of
    44 KB per audio-second -- the only collective after the speaker-bundle broadcast, and like it outside the decode loop.
    Ranks may own different numbers of rows (blocks are padded to the largest shard).  `shards[r]` = the utterance indices of rank r:
    `shard_utterances` is deterministic, so every rank computes the same table and nothing but the waveforms crosses the wire
    (default: the `rank::world` split).  Runs whenever a process group exists (world size 1 included)."""
    if not _group_active():
        out = torch.empty((n_utts,) + tuple(wav.shape[1:]), dtype=wav.dtype, device=wav.device)
        out[torch.as_tensor(list(indices), dtype=torch.long, device=wav.device)] = wav
        return out
    w = world()
    dev = wav.device
    r = rank()
    if shards is None:
        shards = [shard_utterances(n_utts, k, w) for k in range(w)]
    if list(shards[r]) != list(indices) or wav.shape[0] != len(indices):
        raise ValueError("gather_waveform_tensor: `indices` / the block's rows do not match this rank's entry of `shards`")
    cmax = max(len(sh) for sh in shards)
    block = torch.zeros((cmax,) + tuple(wav.shape[1:]), dtype=wav.dtype, device=dev)
    block[: wav.shape[0]] = wav
    # bytes on the wire: neither RCCL nor gloo carries int16
    wire = block.view(torch.uint8)
    parts_b = [torch.empty_like(wire) for _ in range(w)] if r == dst else None
    dist.gather(wire, parts_b, dst=dst)
    if r != dst:
        return None
    out = torch.empty((n_utts,) + tuple(wav.shape[1:]), dtype=wav.dtype, device=dev)
    for sh, part in zip(shards, parts_b):
        if len(sh):
            out[torch.as_tensor(list(sh), dtype=torch.long, device=dev)] = part.view(wav.dtype)[: len(sh)]
    return out
