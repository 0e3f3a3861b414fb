"""oracle/torch_cpu.py -- fp32 PyTorch-CPU (oneDNN) training step of the L3 AVC graph.

TEST INFRASTRUCTURE ONLY, like the rest of `oracle/`: it is the "CPU reference stand-in"
of SURVEY.md 8(d) item (ii) -- the reference's Keras-2.0.9 / TF-1.4 CPU path is not
installable here, and an oneDNN-backed graph is a *stronger* CPU baseline than TF-1.4's
Eigen kernels, so a GPU/CPU ratio quoted against it is conservative.  Only
`bench.py`'s `cpu_baseline` leg (and tests) may import it; the product never does.

Same graph as `oracle/l3_oracle.py` (layer ledger from `model_spec`, which follows
l3embedding/audio_model.py:335-442, vision_model.py:102-195, model.py:7-35) and the same
step semantics (train.py:269-284: categorical cross-entropy with Keras' 1e-7 clip, L2 1e-5
on the 18 kernels, Adam with Keras' lr_t), but convolutions, batch-norm, pooling and all
gradients come from torch (autograd), in float32, with torch's default thread count.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import l3_oracle as o


def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _frontend(kind, audio, freq2mel):
    cfg = o.FRONTENDS[kind]
    x = audio[:, 0, :]
    n_dft, hop = cfg['n_dft'], cfg['n_hop']
    if cfg['padding'] == 'same':
        pl, pr = _same_pad(x.shape[1], n_dft, hop)
        x = F.pad(x, (pl, pr))
    fr = x.unfold(1, n_dft, hop)
    win = torch.hann_window(n_dft, periodic=True, dtype=torch.float32)
    spec = torch.fft.rfft(fr * win, dim=-1)
    p = spec.real ** 2 + spec.imag ** 2
    if cfg['n_mels']:
        p = p @ freq2mel
    if cfg['power'] != 2.0:
        p = torch.sqrt(p) ** cfg['power']
    out = p.permute(0, 2, 1).unsqueeze(-1)
    if cfg['db']:
        ls = 10.0 * torch.log(torch.clamp(out, min=1e-10)) / math.log(10.0)
        mx = ls.reshape(ls.shape[0], -1).max(dim=1).values.reshape(-1, 1, 1, 1)
        out = torch.clamp(ls - mx, min=-80.0)
    if cfg.get('loglambda'):
        out = torch.log(torch.clamp(out, min=1e-12)) / 5.0
    return out


def _tower(prefix, ops, x, T):
    x = x.permute(0, 3, 1, 2).contiguous()
    for op in ops:
        if op[0] == 'conv':
            name, padding = op[1], op[5]
            w = T['%s/%s/kernel' % (prefix, name)].permute(3, 2, 0, 1)
            b = T['%s/%s/bias' % (prefix, name)]
            if padding == 'same':
                pt, pb = _same_pad(x.shape[2], op[3], 1)
                pl, pr = _same_pad(x.shape[3], op[4], 1)
                x = F.pad(x, (pl, pr, pt, pb))
            x = F.conv2d(x, w, b)
        elif op[0] == 'bn':
            name = op[1]
            x = F.batch_norm(x, None, None, T['%s/%s/gamma' % (prefix, name)], T['%s/%s/beta' % (prefix, name)],
                             True, 0.0, o.BN_EPS)
        elif op[0] == 'relu':
            x = F.relu(x)
        elif op[0] == 'pool':
            _, ph, pw, sh, sw, padding = op
            if padding == 'same':
                pt, pb = _same_pad(x.shape[2], ph, sh)
                pl, pr = _same_pad(x.shape[3], pw, sw)
                x = F.pad(x, (pl, pr, pt, pb), value=float('-inf'))
            x = F.max_pool2d(x, (ph, pw), (sh, sw))
        elif op[0] == 'flatten':
            x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    return x


class TorchCpuTrainer(object):
    """Holds fp32 parameters + Adam slots; `step()` = forward + backward + Adam (one training step)."""

    def __init__(self, model_type, P):
        self.model_type = model_type
        self.spec = o.model_spec(model_type)
        self.table = o.param_table(model_type)
        self.T = {}
        for name, _, trainable, _ in self.table:
            t = torch.tensor(np.asarray(P[name], dtype=np.float32))
            t.requires_grad_(bool(trainable))
            self.T[name] = t
        self.m = {n: torch.zeros_like(self.T[n]) for n, _, tr, _ in self.table if tr}
        self.v = {n: torch.zeros_like(self.T[n]) for n, _, tr, _ in self.table if tr}
        self.t = 0

    def step(self, video, audio, labels, lr):
        T = self.T
        fe_key = 'audio_model/%s/freq2mel' % self.spec['frontend_name']
        with torch.no_grad():
            fe = _frontend(self.spec['frontend'], torch.as_tensor(audio, dtype=torch.float32), T.get(fe_key))
        v = _tower('vision_model', self.spec['vision'], torch.as_tensor(video, dtype=torch.float32), T)
        a = _tower('audio_model', self.spec['audio'], fe, T)
        h1 = F.relu(torch.cat([v, a], dim=1) @ T['dense_1/kernel'] + T['dense_1/bias'])
        p = torch.softmax(h1 @ T['dense_2/kernel'] + T['dense_2/bias'], dim=1)
        q = torch.clamp(p / p.sum(dim=1, keepdim=True), o.K_EPSILON, 1 - o.K_EPSILON)
        tl = torch.as_tensor(labels, dtype=torch.float32)
        loss = (-(tl * torch.log(q)).sum(dim=1)).mean()
        loss = loss + sum(o.L2_WEIGHT * (T[n] ** 2).sum() for n, _, _, k in self.table if k == 'kernel')
        for n in self.m:
            T[n].grad = None
        loss.backward()
        self.t += 1
        lr_t = lr * math.sqrt(1.0 - 0.999 ** self.t) / (1.0 - 0.9 ** self.t)
        with torch.no_grad():
            for n in self.m:
                g = T[n].grad
                self.m[n].mul_(0.9).add_(g, alpha=0.1)
                self.v[n].mul_(0.999).addcmul_(g, g, value=0.001)
                T[n].sub_(lr_t * self.m[n] / (self.v[n].sqrt() + 1e-8))
        return float(loss.detach())
