"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32, eval mode) of the ECAPA-TDNN speaker encoder inside the IndexTTS-1 / 1.5
vocoder (`self.speaker_encoder(mel_ref, lens)`, indextts/BigVGAN/models.py:191,202 -- SURVEY.md section 8 row a-13).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows (paths relative to the reference repo root):
  ECAPA_TDNN.forward, TDNNBlock, Res2NetBlock, SEBlock, SERes2NetBlock, AttentiveStatisticsPooling   indextts/BigVGAN/ECAPA_TDNN.py:79-582
  Conv1d ("same" padding in reflect mode, skip_transpose)                                               indextts/BigVGAN/nnet/CNN.py:331-516
  BatchNorm1d (eval: running statistics)                                                                indextts/BigVGAN/nnet/normalization.py
as built by `ECAPA_TDNN(h.num_mels, lin_neurons=h.speaker_embedding_dim)`: channels [C, C, C, C, 3C] (C = 512), kernels [5, 3, 3, 3, 1], dilations
[1, 2, 3, 4, 1], Res2Net scale 8, SE / attention width 128, global context, lengths = None (the pipeline passes none, indextts/infer.py:647).

Pinned by tests/golden/ecapa.npz: outputs of the reference's own class loaded strictly with this module's seeded weights (tools/make_golden_ecapa.py).
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass
class EcapaCfg:
    input_size: int = 100
    lin_neurons: int = 512
    channels: int = 512                 # C; the aggregation layer is 3C wide
    attention_channels: int = 128
    se_channels: int = 128
    res2net_scale: int = 8

KERNELS, DILATIONS = (5, 3, 3, 3, 1), (1, 2, 3, 4, 1)
BN_EPS = 1e-5


def param_shapes(c: EcapaCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    C, S = c.channels, c.res2net_scale
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def tdnn(p, cin, cout, k):
        out.extend([(p + "conv.conv.weight", (cout, cin, k)), (p + "conv.conv.bias", (cout,))])
        bn(p + "norm.norm.", cout)

    def bn(p, ch):
        out.extend([(p + "weight", (ch,)), (p + "bias", (ch,)), (p + "running_mean", (ch,)), (p + "running_var", (ch,)),
                    (p + "num_batches_tracked", ())])

    tdnn("blocks.0.", c.input_size, C, KERNELS[0])
    for i in (1, 2, 3):
        p = f"blocks.{i}."
        tdnn(p + "tdnn1.", C, C, 1)
        for j in range(S - 1):
            tdnn(p + f"res2net_block.blocks.{j}.", C // S, C // S, KERNELS[i])
        tdnn(p + "tdnn2.", C, C, 1)
        out.extend([(p + "se_block.conv1.conv.weight", (c.se_channels, C, 1)), (p + "se_block.conv1.conv.bias", (c.se_channels,)),
                    (p + "se_block.conv2.conv.weight", (C, c.se_channels, 1)), (p + "se_block.conv2.conv.bias", (C,))])
    tdnn("mfa.", 3 * C, 3 * C, 1)
    tdnn("asp.tdnn.", 9 * C, c.attention_channels, 1)
    out.extend([("asp.conv.conv.weight", (3 * C, c.attention_channels, 1)), ("asp.conv.conv.bias", (3 * C,))])
    bn("asp_bn.norm.", 6 * C)
    out.extend([("fc.conv.weight", (c.lin_neurons, 6 * C, 1)), ("fc.conv.bias", (c.lin_neurons,))])
    return out


def synth_weights(c: EcapaCfg, seed: int = 41) -> Dict[str, torch.Tensor]:
    """Seeded weights with non-trivial BatchNorm running statistics under the reference class's names."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(c):
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.tensor(1)
        elif name.endswith("running_var"):
            sd[name] = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith("running_mean"):
            sd[name] = 0.2 * torch.randn(shape, generator=g)
        elif name.endswith("norm.weight"):
            sd[name] = 1 + 0.2 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            sd[name] = 0.1 * torch.randn(shape, generator=g)
        else:
            fan = shape[1] * shape[2]
            sd[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan)
    return sd


def _conv(sd, p, x, dilation=1):
    w = sd[p + "conv.weight"]
    pad = dilation * (w.shape[2] - 1) // 2                          # get_padding_elem, stride 1
    if pad:
        x = F.pad(x, (pad, pad), mode="reflect")
    return F.conv1d(x, w, sd[p + "conv.bias"], dilation=dilation)


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, BN_EPS)


def _tdnn(sd, p, x, dilation=1):
    return _bn(sd, p + "norm.norm.", F.relu(_conv(sd, p + "conv.", x, dilation)))


def _stats(x, m, eps=1e-12):
    mean = (m * x).sum(2)
    return mean, torch.sqrt((m * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(eps))


def ecapa(sd, c: EcapaCfg, feats: torch.Tensor) -> torch.Tensor:
    """feats (B, T, input_size) -> (B, 1, lin_neurons)   (ECAPA_TDNN.forward with lengths=None)"""
    x = feats.transpose(1, 2)
    xl = []
    x = _tdnn(sd, "blocks.0.", x, DILATIONS[0])
    xl.append(x)
    for i in (1, 2, 3):
        p = f"blocks.{i}."
        res = x
        x = _tdnn(sd, p + "tdnn1.", x)
        ys = []
        for j, xj in enumerate(torch.chunk(x, c.res2net_scale, dim=1)):
            if j == 0:
                yj = xj
            elif j == 1:
                yj = _tdnn(sd, p + "res2net_block.blocks.0.", xj, DILATIONS[i])
            else:
                yj = _tdnn(sd, p + f"res2net_block.blocks.{j - 1}.", xj + yj, DILATIONS[i])
            ys.append(yj)
        x = _tdnn(sd, p + "tdnn2.", torch.cat(ys, 1))
        s = x.mean(dim=2, keepdim=True)
        s = torch.sigmoid(_conv(sd, p + "se_block.conv2.", F.relu(_conv(sd, p + "se_block.conv1.", s))))
        x = s * x + res
        xl.append(x)
    x = _tdnn(sd, "mfa.", torch.cat(xl[1:], 1))
    L = x.shape[2]
    mean, std = _stats(x, torch.full((1, 1, L), 1.0 / L))
    attn = torch.cat([x, mean.unsqueeze(2).repeat(1, 1, L), std.unsqueeze(2).repeat(1, 1, L)], 1)
    attn = _conv(sd, "asp.conv.", torch.tanh(_tdnn(sd, "asp.tdnn.", attn)))
    mean, std = _stats(x, F.softmax(attn, dim=2))
    x = _bn(sd, "asp_bn.norm.", torch.cat([mean, std], 1).unsqueeze(2))
    return _conv(sd, "fc.", x).transpose(1, 2)
