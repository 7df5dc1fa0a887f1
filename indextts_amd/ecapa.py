"""ECAPA-TDNN speaker encoder on the HIP engine: the `speaker_encoder` inside the IndexTTS-1 / 1.5 vocoder (`ECAPA_TDNN(h.num_mels,
lin_neurons=h.speaker_embedding_dim)`, indextts/BigVGAN/models.py:191; called on the reference mel at :202) --
host mirror of indextts/BigVGAN/ECAPA_TDNN.py with the reference's parameter names, eval mode, `lengths=None` (how the pipeline calls it).

Once per reference clip, so everything is exact-f32 unit ops of the C ABI on frame-major matrices [T][C]:
  * TDNNBlock = Conv1d ("same" padding in REFLECT mode) -> ReLU -> BatchNorm: the conv is a row gather with reflected indices (torch indexing:
    data movement) + `itts_gemm_forward`; ReLU `itts_tok_act_forward`; the eval-mode BatchNorm FOLLOWS the ReLU, so it is a per-channel affine
    map applied by `itts_tok_affine_forward` (it cannot be folded into the conv).
  * Res2NetBlock: eight channel groups, group j >= 2 convolves (its input + the previous group's output): seven small TDNN blocks in sequence.
  * SEBlock: mean over time (`itts_tok_attnstats_forward` with uniform weights) -> 1x1 -> ReLU -> 1x1 -> sigmoid gate (`itts_tok_gate_forward`).
  * AttentiveStatisticsPooling with global context: [x | mean | std] -> TDNN(1x1) -> tanh -> 1x1 -> per-channel softmax over time -> weighted mean
    and standard deviation, both steps of statistics on `itts_tok_attnstats_forward`.
"""
from typing import Dict, Optional

import torch

from . import _lib
from .campplus import BN_EPS, _COps
from .cond import _Lin

KERNELS, DILATIONS = (5, 3, 3, 3, 1), (1, 2, 3, 4, 1)


class _EOps(_COps):
    def attnstats(self, x, logits=None, eps=1e-12):
        out = torch.empty(1, 2 * x.shape[1], dtype=torch.float32, device=self.device)
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_attnstats_forward(_lib.ptr(x), _lib.ptr(logits), _lib.ptr(out), x.shape[0], x.shape[1], float(eps), self._st()),
                       "itts_tok_attnstats_forward")
        return out


class ECAPA_TDNN:
    """Constructor arguments as the reference class; the fixed structure (five stages, kernels 5 / 3 / 3 / 3 / 1, dilations 1 / 2 / 3 / 4 / 1, ReLU,
    global context, groups 1) is the only one built.  `ops=` is for tests (a stand-in for the C-ABI wrappers)."""

    def __init__(self, input_size: int, device="cuda:0", lin_neurons: int = 192, channels=(512, 512, 512, 512, 1536), kernel_sizes=KERNELS,
                 dilations=DILATIONS, attention_channels: int = 128, res2net_scale: int = 8, se_channels: int = 128, global_context: bool = True,
                 ops=None, **_unused):
        ch = list(channels)
        if tuple(kernel_sizes) != KERNELS or tuple(dilations) != DILATIONS or not global_context or len(ch) != 5 or len(set(ch[:4])) != 1 or \
                ch[4] != 3 * ch[0] or ch[0] % res2net_scale:
            raise NotImplementedError("ECAPA_TDNN (HIP engine): the reference's structure only (channels [C, C, C, C, 3C], kernels 5/3/3/3/1, "
                                      "dilations 1/2/3/4/1, global context)")
        self.input_size, self.lin_neurons, self.C, self.scale = int(input_size), int(lin_neurons), int(ch[0]), int(res2net_scale)
        self.att, self.se = int(attention_channels), int(se_channels)
        self.device = torch.device(device)
        self.ops = ops if ops is not None else _EOps(self.device)
        self._loaded = False

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise _lib.HipEngineError("ECAPA_TDNN handles are bound to their construction device")
        return self

    # ---- weights ---------------------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True, prefix: str = ""):
        """`prefix="speaker_encoder."` reads the encoder out of a v1 / v1.5 BigVGAN checkpoint."""
        dev = self.device
        sd = {k[len(prefix):]: v.detach().cpu().float() for k, v in sd.items() if k.startswith(prefix) and not k.endswith("num_batches_tracked")}

        def conv(p):                                    # Conv1d weight (co, ci, k) -> GEMM over gathered rows, columns (tap, ci)
            w = sd[p + "conv.weight"]
            return _Lin(w.permute(0, 2, 1).reshape(w.shape[0], -1), sd[p + "conv.bias"], dev)

        def bn(p):                                      # eval-mode BatchNorm as (scale, shift)
            s = sd[p + "weight"] / torch.sqrt(sd[p + "running_var"] + BN_EPS)
            return s.to(dev).contiguous(), (sd[p + "bias"] - sd[p + "running_mean"] * s).to(dev).contiguous()

        def tdnn(p):
            return dict(conv=conv(p + "conv."), bn=bn(p + "norm.norm."))

        self.block0 = tdnn("blocks.0.")
        self.blocks = []
        for i in (1, 2, 3):
            p = f"blocks.{i}."
            self.blocks.append(dict(tdnn1=tdnn(p + "tdnn1."), res2=[tdnn(p + f"res2net_block.blocks.{j}.") for j in range(self.scale - 1)],
                                    tdnn2=tdnn(p + "tdnn2."), se1=conv(p + "se_block.conv1."), se2=conv(p + "se_block.conv2."), d=DILATIONS[i],
                                    k=KERNELS[i]))
        self.mfa = tdnn("mfa.")
        self.asp_tdnn, self.asp_conv = tdnn("asp.tdnn."), conv("asp.conv.")
        self.asp_bn = bn("asp_bn.norm.")
        self.fc = conv("fc.")
        if strict and self.block0["conv"].n_out != self.C:
            raise ValueError("ECAPA_TDNN: checkpoint width does not match the constructor's channels")
        self._loaded = True
        return self

    # ---- pieces ------------------------------------------------------------------------------------------------------------------
    def _lin(self, x, lin: _Lin):
        if x.shape[1] != lin.k:
            x = torch.nn.functional.pad(x, (0, lin.k - x.shape[1]))
        return self.ops.linear(x.contiguous(), lin.wp, lin.b, lin.n_out)

    def _tdnn(self, x, blk, k=1, d=1):
        """TDNNBlock on rows: reflect-padded "same" conv (ECAPA's Conv1d default padding mode), ReLU, BatchNorm"""
        T = x.shape[0]
        if k > 1:
            pad = d * (k - 1) // 2
            if pad >= T:
                raise ValueError(f"ECAPA_TDNN: {T} frames are too few for a reflect padding of {pad}")
            t = torch.arange(T, device=x.device)[:, None] + torch.arange(k, device=x.device)[None, :] * d - pad
            t = torch.where(t < 0, -t, t)
            t = torch.where(t >= T, 2 * (T - 1) - t, t)
            x = x[t.reshape(-1)].view(T, k * x.shape[1])
        y = self.ops.act_(self._lin(x, blk["conv"]), 0)
        return self.ops.affine(y, y.shape[1], y.shape[1], *blk["bn"], relu=False)

    def _one(self, feats: torch.Tensor) -> torch.Tensor:
        """feats (T, input_size) -> (1, lin_neurons)"""
        ops, C = self.ops, self.C
        x = feats.to(self.device, torch.float32).contiguous()
        T = x.shape[0]
        x = self._tdnn(x, self.block0, KERNELS[0], DILATIONS[0])
        outs = []
        for blk in self.blocks:                                            # SERes2NetBlock
            res = x
            h = self._tdnn(x, blk["tdnn1"])
            w = C // self.scale
            ys = [h[:, :w]]
            for j in range(1, self.scale):
                xin = h[:, j * w:(j + 1) * w].contiguous()
                if j > 1:
                    xin = ops.add_(xin, ys[-1].contiguous())
                ys.append(self._tdnn(xin, blk["res2"][j - 1], blk["k"], blk["d"]))
            h = self._tdnn(torch.cat(ys, 1).contiguous(), blk["tdnn2"])
            s = ops.attnstats(h)[:, :C].contiguous()                      # mean over time
            g = self._lin(ops.act_(self._lin(s, blk["se1"]), 0), blk["se2"])
            h = ops.gate_(h, g.expand(T, C).contiguous())                 # sigmoid(conv2(relu(conv1(mean)))) * x
            x = ops.add_(h, res)
            outs.append(x)
        x = self._tdnn(torch.cat(outs, 1).contiguous(), self.mfa)          # [T][3C]
        st = ops.attnstats(x)                                              # global context: mean | std with uniform weights
        a = torch.cat([x, st[:, : 3 * C].expand(T, -1), st[:, 3 * C:].expand(T, -1)], 1).contiguous()
        a = ops.act_(self._tdnn(a, self.asp_tdnn), 2)                      # tanh
        logits = self._lin(a, self.asp_conv)
        pooled = ops.attnstats(x, logits)                                  # [1][6C]
        pooled = ops.affine(pooled, 6 * C, 6 * C, *self.asp_bn, relu=False)
        return self._lin(pooled, self.fc)

    def forward(self, x: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x (B, T, input_size) -> (B, 1, lin_neurons); reference clips are encoded one at a time (eval mode: rows are independent)"""
        if not self._loaded:
            raise RuntimeError("ECAPA_TDNN: load_state_dict() first")
        if lengths is not None:
            raise NotImplementedError("ECAPA_TDNN (HIP engine): lengths=None (full-length reference clips), as the pipeline calls it")
        if x.dim() != 3 or x.shape[2] != self.input_size:
            raise ValueError(f"ECAPA_TDNN: expected (B, T, {self.input_size}) features, got {tuple(x.shape)}")
        return torch.stack([self._one(x[b]) for b in range(x.shape[0])], 0)

    __call__ = forward
