"""Developer parity sweep (GPU): HIP ops / full model vs the numpy oracle.  Prints max errors."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import l3_oracle as o
from l3embedding_amd import _lib


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


rng = np.random.RandomState(0)
print('== conv ops')
for (n, h, w, ci, co, k, same) in [(2, 9, 11, 16, 64, 3, 1), (1, 17, 13, 64, 128, 3, 1), (2, 12, 10, 1, 64, 3, 1),
                                   (2, 12, 10, 3, 64, 3, 1), (1, 8, 8, 128, 256, 3, 1), (2, 14, 9, 3, 10, 5, 0),
                                   (2, 14, 9, 10, 10, 5, 0), (3, 7, 5, 32, 48, 3, 1), (1, 6, 6, 256, 512, 3, 1)]:
    x = rng.randn(n, h, w, ci).astype(np.float32)
    wt = (rng.randn(k, k, ci, co) * 0.1).astype(np.float32)
    b = rng.randn(co).astype(np.float32)
    pad = 'same' if same else 'valid'
    y_ref = o.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64), pad)
    y = _lib.op_conv2d_fwd(x, wt, b, same)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dx_ref, dw_ref, db_ref = o.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64), pad)
    dx, dw, db = _lib.op_conv2d_bwd(x, wt, dy, same)
    print((n, h, w, ci, co, k, same), 'fwd %.2e dx %.2e dw %.2e db %.2e' % (relerr(y, y_ref), relerr(dx, dx_ref), relerr(dw, dw_ref), relerr(db, db_ref)))

print('== bn ops')
for (rows, c, relu) in [(1000, 64, 1), (333, 1, 0), (777, 3, 0), (4096, 128, 1), (100, 512, 1), (50, 10, 1)]:
    x = (rng.randn(rows, c) * 3 + 5).astype(np.float32)
    g = rng.rand(c).astype(np.float32) + 0.5; bt = rng.randn(c).astype(np.float32)
    y_ref, cache = o.bn_fwd(x.astype(np.float64), g.astype(np.float64), bt.astype(np.float64), None, None, True)
    if relu: y_ref = np.maximum(y_ref, 0)
    y, mean, var = _lib.op_bn_relu_fwd(x, g, bt, relu)
    dy = rng.randn(rows, c).astype(np.float32)
    dz = np.where(y > 0, dy, 0) if relu else dy
    dx_ref, dg_ref, db_ref = o.bn_bwd(dz.astype(np.float64), g.astype(np.float64), cache, True)
    dx, dg, db = _lib.op_bn_relu_bwd(x, y, dy, g, mean, var, relu)
    print((rows, c, relu), 'y %.2e mean %.2e var %.2e dx %.2e dg %.2e db %.2e' % (relerr(y, y_ref), relerr(mean, cache[2]), relerr(var, cache[3]), relerr(dx, dx_ref), relerr(dg, dg_ref), relerr(db, db_ref)))

print('== pool ops')
for (n, h, w, c, ph, pw, same) in [(2, 8, 8, 64, 2, 2, 0), (2, 9, 7, 16, 2, 2, 0), (2, 9, 7, 16, 2, 2, 1), (2, 32, 24, 8, 32, 24, 0),
                                   (1, 28, 28, 4, 28, 28, 1), (2, 32, 24, 4, 8, 8, 1), (2, 10, 11, 10, 3, 3, 0), (1, 28, 28, 4, 7, 7, 1)]:
    x = rng.randn(n, h, w, c).astype(np.float32)
    y_ref, cache = o.maxpool_fwd(x.astype(np.float64), ph, pw, ph, pw, 'same' if same else 'valid')
    y = _lib.op_maxpool_fwd(x, ph, pw, ph, pw, same)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dx_ref = o.maxpool_bwd(dy.astype(np.float64), cache)
    dx = _lib.op_maxpool_bwd(x, dy, ph, pw, ph, pw, same)
    print((n, h, w, c, ph, pw, same), 'fwd %.2e bwd %.2e' % (relerr(y, y_ref), relerr(dx, dx_ref)))

print('== preprocess')
u8 = np.arange(256, dtype=np.uint8); i16 = np.array([-32768, -1, 0, 1, 32767], np.int16)
vo, ao = _lib.op_preprocess(u8, i16)
print('video exact', np.array_equal(vo, o.preprocess_video(u8)), 'audio exact', np.array_equal(ao, o.pcm2float(i16)))

print('== frontend')
v, a, l = o.synthetic_batch(2)
for mt in ['cnn_L3_melspec2', 'cnn_L3_orig', 'cnn_L3_melspec1', 'cnn_L3_kapredbinputbn', 'tiny_L3']:
    spec = o.model_spec(mt)
    ref = o.frontend_forward(spec['frontend'], a, None, 'sample', np.float64)
    got = _lib.op_frontend(mt, a)
    print(mt, ref.shape, 'max abs err %.3e (range %.1f..%.1f)' % (np.abs(got - ref).max(), ref.min(), ref.max()))
# sine input too (non-noise spectrum)
t = np.arange(48000) / 48000.0
a2 = np.stack([0.5 * np.sin(2 * np.pi * 440 * t), 0.1 * np.sin(2 * np.pi * 3000 * t) + 0.01 * np.sin(2 * np.pi * 100 * t)])[:, None, :].astype(np.float32)
ref = o.frontend_forward('melspec2', a2, None, 'sample', np.float64); got = _lib.op_frontend('cnn_L3_melspec2', a2)
print('sine melspec2 max abs err %.3e ; relative-amplitude err %.3e' % (np.abs(got - ref).max(), np.abs(10 ** (got / 10) - 10 ** (ref / 10)).max()))

print('== full model')
for mt in (sys.argv[1:] or ['cnn_L3_melspec2']):
    B = 2
    P = o.init_params(mt, seed=1)
    # non-trivial BN params so errors are visible
    r2 = np.random.RandomState(5)
    for k in P:
        if k.endswith('/gamma'): P[k] = (1 + 0.1 * r2.randn(*P[k].shape)).astype(np.float32)
        if k.endswith('/beta') or k.endswith('/bias'): P[k] = (0.1 * r2.randn(*P[k].shape)).astype(np.float32)
    v, a, l = o.synthetic_batch(B, seed=3)
    eng = _lib.Engine(mt, B)
    names = [n for n, _, _ in eng.param_table()]
    assert names == [n for n, _, _, _ in o.param_table(mt)], 'param order mismatch'
    for n, s, _ in eng.param_table():
        if '/real_kernels' in n or '/imag_kernels' in n or '/freq2mel' in n:
            print(n, 'const err %.2e' % np.abs(eng.get_param(n, s) - P[n]).max())
    eng.set_params(P)
    t0 = time.time(); out, grads = o.loss_and_grads(mt, P, v, a, l, True, np.float64); print('oracle %.1fs' % (time.time() - t0))
    probs, logits = eng.forward(v, a, training=True)
    print(mt, 'train-mode logits err', np.abs(logits - out['logits']).max(), 'logits', out['logits'].ravel())
    out_e = o.forward(mt, P, v, a, False, np.float64)
    probs_e, logits_e = eng.forward(v, a, training=False)
    print(mt, 'eval-mode logits err', np.abs(logits_e - out_e['logits']).max())
    eng.upload_batch(v, a, l); eng.step_forward(True)
    for b in range(1, eng.bucket_count()): eng.step_backward_bucket(b)
    loss, acc, pr, lg = eng.step_results(True)
    print('loss', loss, 'oracle', out['loss'], 'acc', acc, out['acc'])
    g = eng.get_grads()
    worst = []
    for n in grads:
        gref = grads[n] - (2e-5 * P[n].astype(np.float64) if n.endswith('/kernel') else 0)   # engine adds L2 in Adam
        scale = np.abs(gref).max() + 1e-12
        worst.append((float(np.abs(g[n] - gref).max() / scale), n, float(scale)))
    worst.sort(reverse=True)
    worst = [w_ for w_ in worst if w_[2] > 1e-9]
    for w_ in worst[:12]: print('  grad relerr %.3e %s (scale %.2e)' % w_)
    eng.close()
