"""The RCCL path on the hardware there is: `torch.distributed` with backend "nccl" (= RCCL on ROCm) at world size 1 on the device -- the
process-group initialisation, the flat speaker-bundle broadcast and the uint8-view waveform gather of indextts_amd/dist.py run through the same
RCCL calls as at world size 8 (the multi-rank behaviour is covered by the gloo tests in test_dist.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture()
def nccl_group():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        yield dev
    finally:
        dist.destroy_process_group()


def test_nccl_world1_bundle_broadcast_and_waveform_gather(nccl_group):
    from indextts_amd import dist as D
    dev = nccl_group
    assert dist.get_backend() == "nccl" and D.world() == 1
    g = torch.Generator().manual_seed(3)
    # a full-size bundle: 15 s of reference audio (infer_v2.py:466-470) = 750-frame semantic / emotion sequences, a 1292-frame mel and prompt condition
    bundle = {"style": torch.randn(1, 192, generator=g), "emo_vec": torch.randn(1, 1280, generator=g),
              "spk_cond_emb": torch.randn(1, 750, 1024, generator=g), "emo_cond_emb": torch.randn(1, 750, 1024, generator=g),
              "ref_mel": torch.randn(1, 80, 1292, generator=g), "prompt_condition": torch.randn(1, 1292, 512, generator=g)}
    n_bytes = 4 * sum(t.numel() for t in bundle.values())
    assert 9.0e6 < n_bytes < D.BUNDLE_CAPACITY
    on_dev = {k: v.to(dev) for k, v in bundle.items()}
    got = D.broadcast_speaker_bundle(on_dev, src=0, device=dev)
    torch.cuda.synchronize()
    assert sorted(got) == sorted(bundle)
    for k, v in bundle.items():
        assert got[k].device.type == "cuda" and got[k].shape == v.shape and torch.equal(got[k].cpu(), v), k
    # absent keys stay absent; the payload that follows them is not shifted
    part = D.broadcast_speaker_bundle({"style": on_dev["style"], "ref_mel": on_dev["ref_mel"]}, src=0, device=dev)
    assert sorted(part) == ["ref_mel", "style"] and torch.equal(part["ref_mel"].cpu(), bundle["ref_mel"])
    with pytest.raises(ValueError):
        D.broadcast_speaker_bundle(on_dev, src=0, device=dev, capacity=1 << 20)
    # one rank's share of the 8-GPU point: 8 utterances x 22.4 s at 22.05 kHz, int16, device-resident
    wav = torch.randint(-32768, 32767, (8, 493056), generator=g, dtype=torch.int16).to(dev)
    idx = [5, 0, 3, 7, 1, 2, 6, 4]
    out = D.gather_waveform_tensor(wav, idx, 8, dst=0, shards=[idx])
    torch.cuda.synchronize()
    assert out.device.type == "cuda" and out.dtype == torch.int16 and out.shape == (8, 493056)
    assert torch.equal(out[torch.tensor(idx, device=dev)], wav)
    with pytest.raises(ValueError):
        D.gather_waveform_tensor(wav, idx, 8, dst=0, shards=[list(range(8))])
    t = torch.ones(4, device=dev)
    dist.all_reduce(t)
    dist.barrier()
    assert float(t.sum()) == 4.0
