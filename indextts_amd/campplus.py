"""CAMPPlus speaker encoder on the HIP engine: host mirror of the reference's
`CAMPPlus(feat_dim=80, embedding_size=192)` (indextts/s2mel/modules/campplus/DTDNN.py, layers.py; constructed at
indextts/infer_v2_5.py:218, called at :649 on the mean-normalised 80-bin fbank of the 16 kHz prompt) with the reference's
parameter names.  Produces the `style` vector of the speaker bundle.

Once per speaker prompt, so everything is exact-f32 unit ops of the C ABI on token-major matrices; eval-mode BatchNorm is a
per-channel affine map: folded into the conv it follows (head convs, TDNN, the bottleneck of every dense layer, the final dense),
applied by `itts_tok_affine_forward` where it PRECEDES a conv (the pre-activation `batchnorm-relu` of the dense / transit layers).
Convolutions are row gathers (torch indexing: data movement only) + `itts_gemm_forward`:
  * FCM head (2-D convs over (frequency, time), DTDNN.py:12-48): rows are (f, t) positions, [F * T][C]; a 3 x 3 conv gathers the nine
    neighbour rows (a zero row stands for the padding), stride (2, 1) keeps every other frequency row.
  * D-TDNN: rows are frames.  Dense layers append their 32 output channels to one [T'][C_max] matrix in place (the reference's
    torch.cat chain, layers.py CAMDenseTDNNBlock.forward); the CAM gate = context pooling (`itts_tok_ctxpool_forward`: global mean +
    100-frame segment mean) -> 1x1 -> ReLU -> 1x1 -> sigmoid-multiply (`itts_tok_gate_forward`).
  * StatsPool on `itts_tok_statspool_forward` (mean | unbiased std), then the 1024 -> 192 dense with its non-affine BatchNorm folded.
"""
from typing import Dict

import torch

from . import _lib
from .cond import _Lin, _Ops

BLOCKS = ((12, 3, 1), (24, 3, 2), (16, 3, 2))      # (layers, kernel, dilation), DTDNN.py:78
BN_EPS = 1e-5


class _COps(_Ops):
    def affine(self, x, ld_x, C, scale, shift, relu=True):
        out = torch.empty(x.shape[0], C, dtype=torch.float32, device=self.device)
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_affine_forward(_lib.ptr(x), ld_x, _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(out), x.shape[0], C,
                                                      int(relu), self._st()), "itts_tok_affine_forward")
        return out

    def ctxpool(self, h, seg_len=100):
        out = torch.empty_like(h)
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_ctxpool_forward(_lib.ptr(h), _lib.ptr(out), h.shape[0], h.shape[1], seg_len, self._st()),
                       "itts_tok_ctxpool_forward")
        return out

    def gate_(self, y, g):
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_gate_forward(_lib.ptr(y), _lib.ptr(g), y.numel(), self._st()), "itts_tok_gate_forward")
        return y

    def statspool(self, x):
        out = torch.empty(1, 2 * x.shape[1], dtype=torch.float32, device=self.device)
        with _lib.on_device(self.device):
            _lib.check(self.L.itts_tok_statspool_forward(_lib.ptr(x), _lib.ptr(out), x.shape[0], x.shape[1], self._st()), "itts_tok_statspool_forward")
        return out


def _bn_affine(sd, p):
    """eval-mode BatchNorm `p` as (scale, shift) per channel"""
    var, mean = sd[p + ".running_var"].float(), sd[p + ".running_mean"].float()
    s = 1.0 / torch.sqrt(var + BN_EPS)
    if p + ".weight" in sd:
        s = s * sd[p + ".weight"].float()
    t = -mean * s
    if p + ".bias" in sd:
        t = t + sd[p + ".bias"].float()
    return s, t


class CAMPPlus:
    def __init__(self, feat_dim=80, embedding_size=512, growth_rate=32, bn_size=4, init_channels=128, config_str="batchnorm-relu",
                 memory_efficient=True, device="cuda:0"):
        if (feat_dim, growth_rate, bn_size, init_channels, config_str) != (80, 32, 4, 128, "batchnorm-relu"):
            raise NotImplementedError("CAMPPlus (HIP engine): the shipped configuration only (feat_dim 80, growth 32, bn_size 4, init 128)")
        self.feat_dim, self.emb = feat_dim, embedding_size
        self.device = torch.device(device)
        self.ops = _COps(self.device)
        self._loaded = False

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise _lib.HipEngineError("CAMPPlus handles are bound to their construction device")
        return self

    # ---- weights ---------------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        dev = self.device
        sd = {k: v.detach().cpu() for k, v in sd.items()}

        def conv_bn(wkey, bnkey):                      # conv (no bias) followed by BatchNorm -> one GEMM with bias; taps-major columns
            w = sd[wkey].float()
            s, t = _bn_affine(sd, bnkey)
            w = w * s.view(-1, *([1] * (w.dim() - 1)))
            if w.dim() == 4:                            # (co, ci, i, j) -> columns (i, j, ci)
                w = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
            else:                                       # (co, ci, k) -> columns (k, ci)
                w = w.permute(0, 2, 1).reshape(w.shape[0], -1)
            return _Lin(w, t, dev)

        def pre_bn(bnkey):
            s, t = _bn_affine(sd, bnkey)
            return s.to(dev).contiguous(), t.to(dev).contiguous()

        self.h_conv1 = conv_bn("head.conv1.weight", "head.bn1")
        self.h_blocks = []
        for layer in ("layer1", "layer2"):
            for b in range(2):
                p = f"head.{layer}.{b}."
                self.h_blocks.append(dict(stride=2 if b == 0 else 1, c1=conv_bn(p + "conv1.weight", p + "bn1"), c2=conv_bn(p + "conv2.weight", p + "bn2"),
                                          sc=conv_bn(p + "shortcut.0.weight", p + "shortcut.1") if p + "shortcut.0.weight" in sd else None))
        self.h_conv2 = conv_bn("head.conv2.weight", "head.bn2")
        self.tdnn = conv_bn("xvector.tdnn.linear.weight", "xvector.tdnn.nonlinear.batchnorm")
        self.blocks, self.transits = [], []
        for i, (layers, k, d) in enumerate(BLOCKS):
            blk = []
            for j in range(layers):
                p = f"xvector.block{i + 1}.tdnnd{j + 1}."
                wl = sd[p + "cam_layer.linear_local.weight"].float()
                blk.append(dict(pre=pre_bn(p + "nonlinear1.batchnorm"), cin=sd[p + "linear1.weight"].shape[1],
                                l1=conv_bn(p + "linear1.weight", p + "nonlinear2.batchnorm"),
                                local=_Lin(wl.permute(0, 2, 1).reshape(wl.shape[0], -1), None, dev), k=k, d=d,
                                g1=_Lin(sd[p + "cam_layer.linear1.weight"].float().squeeze(-1), sd[p + "cam_layer.linear1.bias"], dev),
                                g2=_Lin(sd[p + "cam_layer.linear2.weight"].float().squeeze(-1), sd[p + "cam_layer.linear2.bias"], dev)))
            self.blocks.append(blk)
            p = f"xvector.transit{i + 1}."
            self.transits.append(dict(pre=pre_bn(p + "nonlinear.batchnorm"), lin=_Lin(sd[p + "linear.weight"].float().squeeze(-1), None, dev)))
        self.out_pre = pre_bn("xvector.out_nonlinear.batchnorm")
        s, t = _bn_affine(sd, "xvector.dense.nonlinear.batchnorm")
        self.dense = _Lin(sd["xvector.dense.linear.weight"].float().squeeze(-1) * s[:, None], t, dev)
        self._loaded = True
        return self

    # ---- pieces ----------------------------------------------------------------------------------------------------------
    def _lin(self, x, lin: _Lin):
        if x.shape[1] != lin.k:
            x = torch.nn.functional.pad(x, (0, lin.k - x.shape[1]))
        return self.ops.linear(x.contiguous(), lin.wp, lin.b, lin.n_out)

    def _conv2d(self, x, F_in, T, lin: _Lin, stride_f=1, ksize=3):
        """x [F_in * T][C] (row f * T + t) -> [F_out * T][C_out]; 3 x 3 pad 1 (or 1 x 1 pad 0), stride (stride_f, 1)."""
        dev = self.device
        pad = (ksize - 1) // 2
        F_out = (F_in + 2 * pad - ksize) // stride_f + 1
        f = torch.arange(F_out, device=dev)[:, None, None, None] * stride_f + torch.arange(ksize, device=dev)[None, None, :, None] - pad
        t = torch.arange(T, device=dev)[None, :, None, None] + torch.arange(ksize, device=dev)[None, None, None, :] - pad
        ok = (f >= 0) & (f < F_in) & (t >= 0) & (t < T)
        idx = torch.where(ok, f * T + t, torch.full_like(f * T + t, F_in * T))            # (F_out, T, k, k); OOB -> the zero row
        x_ext = torch.cat([x, torch.zeros(1, x.shape[1], device=dev)], 0)
        col = x_ext[idx.reshape(-1)].view(F_out * T, ksize * ksize * x.shape[1])           # columns (i, j, ci)
        return self._lin(col, lin), F_out

    def _conv1d_rows(self, x, lin: _Lin, k, dilation=1, stride=1, pad=0):
        """x [T][C] -> [T_out][C_out], zero padding"""
        dev, T = self.device, x.shape[0]
        T_out = (T + 2 * pad - dilation * (k - 1) - 1) // stride + 1
        t = torch.arange(T_out, device=dev)[:, None] * stride + torch.arange(k, device=dev)[None, :] * dilation - pad
        idx = torch.where((t >= 0) & (t < T), t, torch.full_like(t, T))
        x_ext = torch.cat([x, torch.zeros(1, x.shape[1], device=dev)], 0)
        return self._lin(x_ext[idx.reshape(-1)].view(T_out, k * x.shape[1]), lin)

    def _one(self, feats: torch.Tensor) -> torch.Tensor:
        """feats (T, 80) -> (1, embedding_size)"""
        ops, dev = self.ops, self.device
        T, F = feats.shape[0], self.feat_dim
        x = feats.to(dev, torch.float32).t().contiguous().view(F * T, 1)                  # rows (f, t), one channel
        x, Fc = self._conv2d(x, F, T, self.h_conv1)
        x = ops.act_(x, 0)
        for blk in self.h_blocks:
            out, F2 = self._conv2d(x, Fc, T, blk["c1"], stride_f=blk["stride"])
            out = ops.act_(out, 0)
            out, _ = self._conv2d(out, F2, T, blk["c2"])
            sc = x if blk["sc"] is None else self._conv2d(x, Fc, T, blk["sc"], stride_f=blk["stride"], ksize=1)[0]
            x, Fc = ops.act_(ops.add_(out, sc), 0), F2
        x, Fc = self._conv2d(x, Fc, T, self.h_conv2, stride_f=2)
        x = ops.act_(x, 0)
        C = x.shape[1]
        x = x.view(Fc, T, C).permute(1, 2, 0).reshape(T, C * Fc).contiguous()            # (B, C * F, T) channel order c * F + f
        x = ops.act_(self._conv1d_rows(x, self.tdnn, 5, stride=2, pad=2), 0)             # [T'][128]
        n = x.shape[0]
        for blk, tr in zip(self.blocks, self.transits):
            c_end = blk[0]["cin"] + 32 * len(blk)
            X = torch.empty(n, c_end, dtype=torch.float32, device=dev)
            X[:, : x.shape[1]] = x
            for L in blk:
                cin = L["cin"]
                a = ops.affine(X, c_end, cin, *L["pre"])                                  # BN + ReLU on the first cin columns
                h = ops.act_(self._lin(a, L["l1"]), 0)                                    # 1x1 bottleneck with its BN folded, ReLU
                y = self._conv1d_rows(h, L["local"], L["k"], dilation=L["d"], pad=(L["k"] - 1) // 2 * L["d"])
                g = self._lin(ops.act_(self._lin(ops.ctxpool(h), L["g1"]), 0), L["g2"])
                X[:, cin: cin + 32] = ops.gate_(y, g)
            x = self._lin(ops.affine(X, c_end, c_end, *tr["pre"]), tr["lin"])
        x = ops.affine(x, x.shape[1], x.shape[1], *self.out_pre)
        return self._lin(ops.statspool(x), self.dense)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, T, 80) -> (B, embedding_size)   (CAMPPlus.forward, DTDNN.py:110-115); prompts are encoded one at a time"""
        if not self._loaded:
            raise RuntimeError("CAMPPlus: load_state_dict() first")
        if x.dim() != 3 or x.shape[2] != self.feat_dim or x.shape[1] < 4:
            raise ValueError(f"CAMPPlus: expected (B, T >= 4, {self.feat_dim}) features, got {tuple(x.shape)}")
        return torch.cat([self._one(x[b]) for b in range(x.shape[0])], 0)

    __call__ = forward
