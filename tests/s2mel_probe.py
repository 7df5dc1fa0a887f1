"""Helper for test_gpu_s2mel.py::test_tile_gemm_kernels_agree / test_f32_fast_path_vs_separate_kernels: runs the s2mel solve (PROBE_PREC,
default bf16) at the shipped widths on three ragged utterances and prints a digest of the raw output (PROBE_SAVE: also saves it).  PROBE_OPTS
("name=value,...": engine options applied through itts_set_option before the model is built, e.g. tile256 = 0 / 1 / 2: the 128 x 128, 256 x 256
eight-wave, 256 x 128 four-wave tile GEMM for every shape) selects the kernels under test, one process per setting."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import s2mel_oracle as S  # noqa: E402  (seeded synthetic weights only)
from test_gpu_s2mel import engine  # noqa: E402
from indextts_amd import _lib  # noqa: E402
for _kv in filter(None, os.environ.get("PROBE_OPTS", "").split(",")):      # engine options of this run (itts_set_option), e.g. "decode_fuse_ln=0"
    _lib.set_option(_kv.split("=")[0], int(_kv.split("=")[1]))

cfg = S.S2MelConfig(depth=3, wavenet_layers=2, wavenet_dilation_rate=int(os.environ.get("PROBE_DIL", "1")))
sd = S.synth_weights(cfg, 5)
m = engine(cfg, sd, os.environ.get("PROBE_PREC", "bf16"))
g = torch.Generator().manual_seed(6)
T, Tp = [391, 97, 258], [40, 33, 1]          # 2 x 746 rows: three 256-row tiles, the last one ragged; sequences straddle tiles
Tm = max(T)
x = torch.randn(3, 80, Tm, generator=g)
mu = torch.randn(3, Tm, cfg.content_dim, generator=g)
prompt = torch.randn(3, 80, max(Tp), generator=g) * 0.5 - 1.0
style = torch.randn(3, cfg.style_dim, generator=g)
h = hashlib.sha256()
for rep in range(int(os.environ.get("PROBE_REPS", "3"))):          # repeated: a race between LDS-DMA and fragment reads is intermittent
    y = m.solve_euler(x.clone(), torch.tensor(T), prompt, mu, style, None, torch.linspace(0, 1, 3), 0.7, prompt_lens=Tp, frame_lens=T)
    h.update(y.float().cpu().numpy().tobytes())
if os.environ.get("PROBE_SAVE"):
    torch.save(y.float().cpu(), os.environ["PROBE_SAVE"])
print("DIGEST", h.hexdigest(), float(y.float().abs().mean()), tuple(y.shape))
