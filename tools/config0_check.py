"""GPU box: the configs[0] extra of bench.py alone (IndexTTS-1.5, one utterance, greedy, v1 vocoder)."""
import json
import os
import sys
import types

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

for prec in ("bf16", "fp32"):
    args = types.SimpleNamespace(precision=prec)
    print(prec, json.dumps(bench.config0_leg(args, torch.device("cuda:0"))), flush=True)
