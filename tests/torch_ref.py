"""Independent PyTorch-CPU (autograd) build of the L3 AVC graph, used only to
cross-check the numpy oracle's hand-written forward/backward.
It shares the layer ledger (`oracle.model_spec`) but none of the arithmetic: convs,
batch-norm, pooling, softmax and gradients come from torch; the STFT from torch.fft."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import l3_oracle as o


def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def frontend_torch(kind, audio, freq2mel=None, db_max_scope='sample'):
    cfg = o.FRONTENDS[kind]
    x = torch.as_tensor(audio, dtype=torch.float64)[:, 0, :]
    n_dft, hop = cfg['n_dft'], cfg['n_hop']
    if cfg['padding'] == 'same':
        pl, pr = _same_pad(x.shape[1], n_dft, hop)
        x = F.pad(x, (pl, pr))
    fr = x.unfold(1, n_dft, hop)                                   # (B, frames, n_dft)
    win = torch.hann_window(n_dft, periodic=True, dtype=torch.float64)
    win = win.to(torch.float32).to(torch.float64)                  # kapre casts the window to floatx
    spec = torch.fft.rfft(fr * win, dim=-1)
    p = spec.real ** 2 + spec.imag ** 2                            # (B, frames, freq)
    if cfg['n_mels']:
        p = p @ torch.as_tensor(freq2mel, dtype=torch.float64)
    if cfg['power'] != 2.0:
        p = torch.sqrt(p) ** cfg['power']
    out = p.permute(0, 2, 1).unsqueeze(-1)
    if cfg['db']:
        ls = 10.0 * torch.log(torch.clamp(out, min=1e-10)) / math.log(10.0)
        if db_max_scope == 'sample':
            mx = ls.reshape(ls.shape[0], -1).max(dim=1).values.reshape(-1, 1, 1, 1)
        else:
            mx = ls.max()
        out = torch.clamp(ls - mx, min=-80.0)
    if cfg['loglambda']:
        out = torch.log(torch.clamp(out, min=1e-12)) / 5.0
    return out


def _tower(prefix, ops, x, T, training, taps=None):
    # x NHWC -> NCHW
    x = x.permute(0, 3, 1, 2)
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
            if taps is not None:
                taps[name] = x.permute(0, 2, 3, 1)
        elif op[0] == 'bn':
            name = op[1]
            g, bt = T['%s/%s/gamma' % (prefix, name)], T['%s/%s/beta' % (prefix, name)]
            mm, mv = T['%s/%s/moving_mean' % (prefix, name)], T['%s/%s/moving_variance' % (prefix, name)]
            if training:
                x = F.batch_norm(x, None, None, g, bt, True, 0.0, o.BN_EPS)
            else:
                x = F.batch_norm(x, mm, mv, g, bt, False, 0.0, o.BN_EPS)
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


def loss_and_grads_torch(model_type, P, video, audio, labels, training=True, db_max_scope='sample'):
    spec = o.model_spec(model_type)
    T = {}
    for name, _, trainable, _ in o.param_table(model_type):
        t = torch.tensor(np.asarray(P[name], dtype=np.float64))
        t.requires_grad_(trainable)
        T[name] = t
    fe_key = 'audio_model/%s/freq2mel' % spec['frontend_name']
    fe = frontend_torch(spec['frontend'], audio, P.get(fe_key), db_max_scope)
    v = _tower('vision_model', spec['vision'], torch.as_tensor(video, dtype=torch.float64), T, training)
    a = _tower('audio_model', spec['audio'], fe, T, training)
    h0 = torch.cat([v, a], dim=1)
    h1 = F.relu(h0 @ T['dense_1/kernel'] + T['dense_1/bias'])
    logits = h1 @ T['dense_2/kernel'] + T['dense_2/bias']
    p = torch.softmax(logits, dim=1)
    t = torch.as_tensor(labels, dtype=torch.float64)
    q = p / p.sum(dim=1, keepdim=True)
    q = torch.clamp(q, o.K_EPSILON, 1 - o.K_EPSILON)
    data_loss = (-(t * torch.log(q)).sum(dim=1)).mean()
    reg = sum(o.L2_WEIGHT * (T[n] ** 2).sum() for n, _, _, k in o.param_table(model_type) if k == 'kernel')
    loss = data_loss + reg
    loss.backward()
    grads = {n: T[n].grad.numpy() for n, _, tr, _ in o.param_table(model_type) if tr}
    return dict(loss=float(loss), logits=logits.detach().numpy(), probs=p.detach().numpy(),
                frontend=fe.numpy()), grads
