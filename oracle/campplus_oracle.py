"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32, eval mode) of the CAMPPlus speaker encoder that produces the `style`
vector of the speaker bundle (indextts/infer_v2_5.py:218,643-649).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.

Follows (paths relative to the reference repo root):
  CAMPPlus.forward / FCM.forward           indextts/s2mel/modules/campplus/DTDNN.py:12-115
  BasicResBlock, TDNNLayer, CAMLayer (+ seg_pooling), CAMDenseTDNNLayer / Block, TransitLayer, DenseLayer, StatsPool,
  get_nonlinear ('batchnorm-relu', 'batchnorm_')                                indextts/s2mel/modules/campplus/layers.py
BatchNorm layers are in eval mode (running statistics), as in the pipeline (`campplus_model.eval()`, infer_v2_5.py:221).

Pinned by tests/golden/campplus.npz: outputs of the reference's own `CAMPPlus` class loaded with this module's seeded weights
(tools/make_golden_campplus.py).
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F

BLOCKS = ((12, 3, 1), (24, 3, 2), (16, 3, 2))          # (layers, kernel, dilation) of the three dense blocks (DTDNN.py:78)
GROWTH, BN_SIZE, INIT_CH, M_CH, FEAT, EMB = 32, 4, 128, 32, 80, 192
SEG_LEN = 100


def _bn_keys(sd, g, p, c, affine=True):
    if affine:
        sd[p + ".weight"] = 1 + 0.2 * torch.randn(c, generator=g)
        sd[p + ".bias"] = 0.1 * torch.randn(c, generator=g)
    sd[p + ".running_mean"] = 0.2 * torch.randn(c, generator=g)
    sd[p + ".running_var"] = 0.5 + torch.rand(c, generator=g)
    sd[p + ".num_batches_tracked"] = torch.tensor(1)


def synth_weights(seed: int = 17) -> Dict[str, torch.Tensor]:
    """Seeded weights + non-trivial BatchNorm running statistics under the reference CAMPPlus(feat_dim=80, embedding_size=192) names."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, fan: torch.randn(*s, generator=g) * math.sqrt(2.0 / fan)
    sd: Dict[str, torch.Tensor] = {}
    sd["head.conv1.weight"] = rn(M_CH, 1, 3, 3, fan=9)
    _bn_keys(sd, g, "head.bn1", M_CH)
    for layer in ("layer1", "layer2"):
        for b in range(2):
            p = f"head.{layer}.{b}."
            sd[p + "conv1.weight"] = rn(M_CH, M_CH, 3, 3, fan=9 * M_CH)
            _bn_keys(sd, g, p + "bn1", M_CH)
            sd[p + "conv2.weight"] = rn(M_CH, M_CH, 3, 3, fan=9 * M_CH)
            _bn_keys(sd, g, p + "bn2", M_CH)
            if b == 0:                                    # stride 2 block: 1x1 shortcut conv + BN
                sd[p + "shortcut.0.weight"] = rn(M_CH, M_CH, 1, 1, fan=M_CH)
                _bn_keys(sd, g, p + "shortcut.1", M_CH)
    sd["head.conv2.weight"] = rn(M_CH, M_CH, 3, 3, fan=9 * M_CH)
    _bn_keys(sd, g, "head.bn2", M_CH)
    ch = M_CH * (FEAT // 8)
    sd["xvector.tdnn.linear.weight"] = rn(INIT_CH, ch, 5, fan=5 * ch)
    _bn_keys(sd, g, "xvector.tdnn.nonlinear.batchnorm", INIT_CH)
    ch = INIT_CH
    bn_ch = BN_SIZE * GROWTH
    for i, (layers, k, _d) in enumerate(BLOCKS):
        for j in range(layers):
            p = f"xvector.block{i + 1}.tdnnd{j + 1}."
            cin = ch + j * GROWTH
            _bn_keys(sd, g, p + "nonlinear1.batchnorm", cin)
            sd[p + "linear1.weight"] = rn(bn_ch, cin, 1, fan=cin)
            _bn_keys(sd, g, p + "nonlinear2.batchnorm", bn_ch)
            sd[p + "cam_layer.linear_local.weight"] = rn(GROWTH, bn_ch, k, fan=k * bn_ch)
            sd[p + "cam_layer.linear1.weight"] = rn(bn_ch // 2, bn_ch, 1, fan=bn_ch)
            sd[p + "cam_layer.linear1.bias"] = 0.1 * torch.randn(bn_ch // 2, generator=g)
            sd[p + "cam_layer.linear2.weight"] = rn(GROWTH, bn_ch // 2, 1, fan=bn_ch // 2)
            sd[p + "cam_layer.linear2.bias"] = 0.1 * torch.randn(GROWTH, generator=g)
        ch = ch + layers * GROWTH
        _bn_keys(sd, g, f"xvector.transit{i + 1}.nonlinear.batchnorm", ch)
        sd[f"xvector.transit{i + 1}.linear.weight"] = rn(ch // 2, ch, 1, fan=ch)
        ch //= 2
    _bn_keys(sd, g, "xvector.out_nonlinear.batchnorm", ch)
    sd["xvector.dense.linear.weight"] = rn(EMB, 2 * ch, 1, fan=2 * ch)
    _bn_keys(sd, g, "xvector.dense.nonlinear.batchnorm", EMB, affine=False)
    return sd


def _bn(sd, p, x, eps=1e-5):
    """eval-mode BatchNorm over the channel dim (dim 1) of a (B, C, ...) tensor"""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd.get(p + ".weight"), sd.get(p + ".bias"), False, 0.0, eps)


def _res_block(sd, p, x, stride):                                  # layers.py BasicResBlock
    out = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"], stride=(stride, 1), padding=1)))
    out = _bn(sd, p + "bn2", F.conv2d(out, sd[p + "conv2.weight"], padding=1))
    sc = x
    if p + "shortcut.0.weight" in sd:
        sc = _bn(sd, p + "shortcut.1", F.conv2d(x, sd[p + "shortcut.0.weight"], stride=(stride, 1)))
    return F.relu(out + sc)


def fcm(sd, x: torch.Tensor) -> torch.Tensor:                       # DTDNN.py:12-48: (B, 80, T) -> (B, 320, T)
    out = F.relu(_bn(sd, "head.bn1", F.conv2d(x.unsqueeze(1), sd["head.conv1.weight"], padding=1)))
    for layer in ("layer1", "layer2"):
        out = _res_block(sd, f"head.{layer}.0.", out, 2)
        out = _res_block(sd, f"head.{layer}.1.", out, 1)
    out = F.relu(_bn(sd, "head.bn2", F.conv2d(out, sd["head.conv2.weight"], stride=(2, 1), padding=1)))
    b, c, f, t = out.shape
    return out.reshape(b, c * f, t)


def seg_pooling(x: torch.Tensor, seg_len: int = SEG_LEN) -> torch.Tensor:       # layers.py CAMLayer.seg_pooling ('avg')
    seg = F.avg_pool1d(x, kernel_size=seg_len, stride=seg_len, ceil_mode=True)
    shape = seg.shape
    seg = seg.unsqueeze(-1).expand(*shape, seg_len).reshape(*shape[:-1], -1)
    return seg[..., : x.shape[-1]]


def cam_dense_layer(sd, p, x, k, d):                                # CAMDenseTDNNLayer.forward + CAMLayer.forward
    h = F.conv1d(F.relu(_bn(sd, p + "nonlinear1.batchnorm", x)), sd[p + "linear1.weight"])
    h = F.relu(_bn(sd, p + "nonlinear2.batchnorm", h))
    y = F.conv1d(h, sd[p + "cam_layer.linear_local.weight"], padding=(k - 1) // 2 * d, dilation=d)
    ctx = h.mean(-1, keepdim=True) + seg_pooling(h)
    ctx = F.relu(F.conv1d(ctx, sd[p + "cam_layer.linear1.weight"], sd[p + "cam_layer.linear1.bias"]))
    m = torch.sigmoid(F.conv1d(ctx, sd[p + "cam_layer.linear2.weight"], sd[p + "cam_layer.linear2.bias"]))
    return y * m


def campplus(sd, feats: torch.Tensor) -> torch.Tensor:
    """feats (B, T, 80) mean-normalised fbank -> (B, 192)   (CAMPPlus.forward, DTDNN.py:110-115)"""
    x = fcm(sd, feats.permute(0, 2, 1))
    x = F.relu(_bn(sd, "xvector.tdnn.nonlinear.batchnorm", F.conv1d(x, sd["xvector.tdnn.linear.weight"], stride=2, padding=2)))
    for i, (layers, k, d) in enumerate(BLOCKS):
        for j in range(layers):
            x = torch.cat([x, cam_dense_layer(sd, f"xvector.block{i + 1}.tdnnd{j + 1}.", x, k, d)], dim=1)
        x = F.conv1d(F.relu(_bn(sd, f"xvector.transit{i + 1}.nonlinear.batchnorm", x)), sd[f"xvector.transit{i + 1}.linear.weight"])
    x = F.relu(_bn(sd, "xvector.out_nonlinear.batchnorm", x))
    stats = torch.cat([x.mean(dim=-1), x.std(dim=-1, unbiased=True)], dim=-1)            # StatsPool
    y = F.conv1d(stats.unsqueeze(-1), sd["xvector.dense.linear.weight"]).squeeze(-1)
    return _bn(sd, "xvector.dense.nonlinear.batchnorm", y)
