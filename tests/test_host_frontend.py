"""CPU: host-only pieces of indextts_amd/frontend.py (WAV decoding conventions; the engine-backed classes are GPU-tested in tests/test_gpu_frontend.py)."""
import numpy as np
import pytest
import torch
from scipy.io import wavfile

from indextts_amd.frontend import EngineFrontend, EngineFrontendV1, _get, _strip_module, load_wav


def test_load_wav_scales_and_mixes_down(tmp_path):
    g = np.random.RandomState(0)
    x = (g.rand(1000, 2) * 2 - 1).astype(np.float32) * 0.9
    cases = {"f32": x, "i16": np.round(x * 32767).astype(np.int16), "i32": np.round(x.astype(np.float64) * (2 ** 31 - 1)).astype(np.int32),
             "u8": np.round((x + 1) * 127.5).astype(np.uint8)}
    tol = {"f32": 0.0, "i16": 2.0 / 32768, "i32": 1e-6, "u8": 2.0 / 127}
    for tag, data in cases.items():
        p = tmp_path / f"{tag}.wav"
        wavfile.write(str(p), 22050, data)
        y, sr = load_wav(str(p))
        assert sr == 22050 and y.shape == (1, 1000) and y.dtype == torch.float32
        assert float((y[0] - torch.from_numpy(x.mean(axis=1))).abs().max()) <= tol[tag] + 1e-7, tag
    wavfile.write(str(tmp_path / "mono.wav"), 16000, cases["i16"][:, 0])
    y, sr = load_wav(str(tmp_path / "mono.wav"))
    assert sr == 16000 and torch.equal(y[0], torch.from_numpy(cases["i16"][:, 0].astype(np.float32) / 32768.0))


def test_config_and_checkpoint_helpers():
    class Obj:
        a = 3
    assert _get({"a": 1}, "a") == 1 and _get({"a": 1}, "b", 7) == 7 and _get(Obj(), "a") == 3 and _get(Obj(), "zz", None) is None
    assert _strip_module({"module.x": 1, "y": 2}) == {"x": 1, "y": 2}


def test_engine_frontends_fail_loudly_without_their_inputs(tmp_path):
    with pytest.raises(FileNotFoundError):
        EngineFrontend({"w2v_stat": "s.pt"}, str(tmp_path), "cpu")                      # no hf_cache/w2v-bert-2.0/config.json
    if not torch.cuda.is_available():
        fe = EngineFrontendV1("cpu")                                                    # constructing is free; the DSP needs the GPU
        wavfile.write(str(tmp_path / "a.wav"), 24000, np.zeros(3000, dtype=np.int16))
        with pytest.raises(Exception):
            fe.cond_mel(str(tmp_path / "a.wav"))
        with pytest.raises(RuntimeError):
            fe.conditioning(torch.zeros(1, 100, 5), torch.tensor([5]))
