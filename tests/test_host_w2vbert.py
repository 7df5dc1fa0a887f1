"""Host logic of indextts_amd/w2vbert.py on CPU: the C-ABI unit ops are replaced by torch stand-ins (TEST ONLY -- the product has no
CPU path) so that the weight folding (q|k|v as one GEMM), the packed-row tables, the relative-key bookkeeping and the layer order
are checked against tests/golden/w2vbert.npz (transformers' Wav2Vec2BertModel) without a GPU.  The kernels themselves are covered
by tests/test_gpu_w2vbert.py."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import w2vbert_oracle as WO
from tools.make_golden_w2vbert import CFG, LAYER


class _FakeLin:
    def __init__(self, w, b, device):
        self.w, self.b, self.n_out, self.k = w.float(), None if b is None else b.float(), w.shape[0], w.shape[1]
        self.wp = self


class _FakeOps:
    device = torch.device("cpu")

    def linear(self, x, wp, bias, n_out):
        return F.linear(x, wp.w, bias)

    def act_(self, x, mode):
        return x.copy_(F.silu(x) if mode == 1 else F.relu(x))

    def glu(self, x, mode):
        assert mode == 0
        return F.glu(x, dim=1)

    def scale_residual_(self, x, y, gamma):
        return x.add_(gamma * y)

    def add_(self, x, y):
        return x.add_(y)

    def dwconv_causal(self, x, w, tok_seq, tok_t, seq_T, k):
        y = torch.zeros_like(x)
        for m in range(x.shape[0]):
            t = int(tok_t[m])
            for j in range(k):
                u = t + j - (k - 1)
                if u >= 0:
                    y[m] += w[:, j] * x[m + u - t]
        return y

    def attention_relkey(self, q, k, v, kstart, klen, qpos, dist, left, right, H, dq, dv, scale):
        n = q.shape[0]
        out = torch.zeros(n, H * dv)
        for m in range(n):
            ks, kl, qp = int(kstart[m]), int(klen[m]), int(qpos[m])
            idx = (torch.arange(kl) - qp).clamp(-left, right) + left
            for h in range(H):
                qq = q[m, h * dq:(h + 1) * dq]
                s = (k[ks:ks + kl, h * dq:(h + 1) * dq] @ qq + dist[idx] @ qq) * scale
                out[m, h * dv:(h + 1) * dv] = torch.softmax(s, 0) @ v[ks:ks + kl, h * dv:(h + 1) * dv]
        return out


def test_host_composition_vs_transformers(golden_dir, monkeypatch):
    from indextts_amd import w2vbert as W
    monkeypatch.setattr(W, "_Lin", _FakeLin)
    monkeypatch.setattr(W, "_WOps", lambda dev: _FakeOps())
    monkeypatch.setattr(W, "layernorm", lambda x, g, b, eps=1e-5: F.layer_norm(x, (x.shape[1],), g, b, eps))
    z = np.load(os.path.join(golden_dir, "w2vbert.npz"))
    m = W.Wav2Vec2BertModel(**CFG.__dict__, device="cpu").load_state_dict(WO.synth_weights(CFG))
    feats, mask = torch.from_numpy(z["feats"]), torch.from_numpy(z["mask"])
    emb = m.get_emb(feats, mask, torch.from_numpy(z["mean"]), torch.from_numpy(z["std"]), layer=LAYER)
    valid = mask.bool()
    assert float((emb - torch.from_numpy(z["emb"]))[valid].abs().max()) <= 2e-5
    out = m(feats, mask, output_hidden_states=True)
    assert len(out.hidden_states) == CFG.num_hidden_layers + 1
    assert float((out.hidden_states[1] - torch.from_numpy(z["h1"]))[valid].abs().max()) <= 2e-5
    assert float((out.last_hidden_state - torch.from_numpy(z["last"]))[valid].abs().max()) <= 2e-5
    assert float(out.last_hidden_state[~valid].abs().max()) == 0.0
    try:
        bad = mask.clone(); bad[0, 3] = 0
        m(feats, bad)
        raise AssertionError("left / interior padding must be rejected")
    except ValueError:
        pass
