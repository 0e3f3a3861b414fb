"""Layer-local parity at the REAL layer geometries of cnn_L3_melspec2.

Every convolution / BatchNorm-ReLU-pool stage of the two towers is fed the same seeded input
the float64 oracle gets, at the layer's real H x W x C (N = 2), so rounding drift cannot
compound across layers and the bound can be tight: max |err| <= 3e-6 of the reference's
range for fp32 forward, data gradient, weight gradient and bias gradient, and the same
for the mixed-precision (bf16 operand / fp32 accumulate) kernels against the oracle run
with the same operand rounding -- in both operand forms the engine uses (fp32 tensors
rounded at fetch, and tensors stored in HBM as bfloat16).
"""
import numpy as np
import pytest

from l3embedding_amd import _lib
from conftest import need_experiments
from oracle import l3_oracle as o

pytestmark = pytest.mark.gpu

# (tag, H, W, Cin, Cout) -- audio_model.py:376-431, vision_model.py:130-184
LEDGER_CONVS = [
    ('A.conv1a', 256, 199, 1, 64), ('A.conv1b', 256, 199, 64, 64), ('A.conv2a', 128, 99, 64, 128),
    ('A.conv2b', 128, 99, 128, 128), ('A.conv3a', 64, 49, 128, 256), ('A.conv3b', 64, 49, 256, 256),
    ('A.conv4a', 32, 24, 256, 512), ('A.conv4b', 32, 24, 512, 512),
    ('V.conv1a', 224, 224, 3, 64), ('V.conv1b', 224, 224, 64, 64), ('V.conv2a', 112, 112, 64, 128),
    ('V.conv2b', 112, 112, 128, 128), ('V.conv3a', 56, 56, 128, 256), ('V.conv3b', 56, 56, 256, 256),
    ('V.conv4a', 28, 28, 256, 512), ('V.conv4b', 28, 28, 512, 512)]
N = 2
TOL = 3e-6          # measured on MI355X: <= 1.2e-6 everywhere (profiles/r02_parity_distances.txt)
# Forward / data gradient of the layers with >= 64 input channels (all 14) run as Winograd F(4x4,3x3) (conv_wino4.hip): 4x fewer
# multiplies than direct, transform coefficients up to 8 instead of 1 -- its own budget.  float32 NumPy restatement of the
# same algorithm: 7-9e-6 of the output range (profiles/r03_wino4_error_model.txt)
TOL_F4 = 3e-5
F4_MIN_CIN = 64


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def _layer_data(tag, h, w, ci, co):
    """What the layer sees in the network: a post-ReLU (or BatchNorm-ed input) activation, a he_normal
    filter, a small bias, and an output gradient of BatchNorm-backward magnitude."""
    rng = np.random.RandomState(sum(map(ord, tag)) + h + ci)
    x = rng.randn(N, h, w, ci).astype(np.float32)
    if ci >= 64:
        x = np.maximum(x, 0)                                  # fed by BN -> ReLU (-> pool)
    wt = (rng.randn(3, 3, ci, co) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = (0.1 * rng.randn(co)).astype(np.float32)
    dy = (rng.randn(N, h, w, co) * 1e-3).astype(np.float32)
    return x, wt, b, dy


@pytest.mark.parametrize('algo', ['product', 'f2x2', 'f2x2_bf16x6'])
@pytest.mark.parametrize('case', LEDGER_CONVS, ids=[c[0] for c in LEDGER_CONVS])
def test_conv_layer_fp32(gpu_required, case, algo, monkeypatch):
    """product: what the engine runs by default (F(4x4,3x3) forward / data gradient for the 14 layers with >= 64 input channels);
    f2x2: those layers on F(2x2,3x3) (L3_WINO4=0), the round-2 configuration and the lower-error alternative;
    f2x2_bf16x6: F(2x2,3x3) with both operands split exactly into three bfloat16 terms, six cross products on the bf16
    matrix cores, fp32 accumulate (conv_wino_bx6.hip) -- held to the SAME bound as the fp32 F(2x2,3x3) kernel."""
    tag, h, w, ci, co = case
    if algo == 'f2x2':
        if max(ci, co) < F4_MIN_CIN:
            pytest.skip('same kernels as the product configuration')
        monkeypatch.setenv('L3_WINO4', '0')
    if algo == 'f2x2_bf16x6':
        need_experiments()
        if max(ci, co) < F4_MIN_CIN:
            pytest.skip('same kernels as the product configuration')
        monkeypatch.setenv('L3_FP32_CONV', 'f2x2_bf16x6')
    x, wt, b, dy = _layer_data(*case)
    x64, w64, b64, dy64 = (t.astype(np.float64) for t in (x, wt, b, dy))
    y_ref = o.conv2d_fwd(x64, w64, b64, 'same')
    dx_ref, dw_ref, db_ref = o.conv2d_bwd(x64, w64, dy64, 'same')
    y = _lib.op_conv2d_fwd(x, wt, b, True)
    dx, dw, db = _lib.op_conv2d_bwd(x, wt, dy, True)
    errs = dict(y=relerr(y, y_ref), dx=relerr(dx, dx_ref), dw=relerr(dw, dw_ref), db=relerr(db, db_ref))
    print(tag, algo, ' '.join('%s=%.2e' % kv for kv in errs.items()))
    f4 = algo == 'product'
    tol = dict(y=TOL_F4 if f4 and ci >= F4_MIN_CIN else TOL, dx=TOL_F4 if f4 and co >= F4_MIN_CIN else TOL, dw=TOL, db=TOL)
    assert all(errs[k] < tol[k] for k in errs), (tag, algo, errs)
    # (which kernel ran is pinned by the engine's own account of issued flops, test_fp32_step_runs_the_winograd_kernels: the
    # error alone cannot tell -- a layer whose tile blocks are split over channel slices (conv_wino4_launch, the tail) sums its
    # slices separately and lands BELOW 1e-6 of the range, where F(2x2,3x3) lives)


@pytest.mark.parametrize('shape', [(16, 56, 56, 256, 256), (32, 28, 28, 512, 128), (8, 112, 112, 64, 64)])
def test_winograd_f4_race_screen(gpu_required, shape):
    """The F(4x4,3x3) kernel keeps LDS-DMA loads in flight across its workgroup barriers behind counted vmcnt waits
    (conv_wino4.hip: the second half's filter slices of the next stage, and the first operand read of the next stage).
    A read that beats its DMA passes single runs whenever the load happens to land first, so: sizes that keep every
    CU busy for several tile blocks, repeated, must agree bit for bit with the first run -- and the first run with
    the float64 oracle on a sample of pixels."""
    n, h, w, ci, co = shape
    rng = np.random.RandomState(ci + h)
    x = np.maximum(rng.randn(n, h, w, ci), 0).astype(np.float32)
    wt = (rng.randn(3, 3, ci, co) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = (0.1 * rng.randn(co)).astype(np.float32)
    y0 = _lib.op_conv2d_fwd(x, wt, b, True)
    for _ in range(6):
        assert np.array_equal(_lib.op_conv2d_fwd(x, wt, b, True), y0)
    ref = o.conv2d_fwd(x[:2].astype(np.float64), wt.astype(np.float64), b.astype(np.float64), 'same')
    assert relerr(y0[:2], ref) < TOL_F4
    # the last sample too (tail tile blocks)
    ref = o.conv2d_fwd(x[-1:].astype(np.float64), wt.astype(np.float64), b.astype(np.float64), 'same')
    assert relerr(y0[-1:], ref) < TOL_F4


@pytest.mark.parametrize('shape', [(3, 13, 18, 16, 64), (2, 9, 7, 24, 128), (5, 30, 21, 72, 64), (4, 16, 16, 40, 192)])
def test_winograd_f4_short_and_odd_k_loops(gpu_required, shape, monkeypatch):
    """The software-pipelined stage loop of conv_wino4.hip at its edges: 2, 3, 5 and 9 eight-channel stages (the product
    configuration starts at 8), ragged tile rows / columns and a partial last tile block -- forward and data gradient
    against the float64 oracle, twice (bit-identical).  L3_WINO4 lowers the kernel's minimum channel count."""
    monkeypatch.setenv('L3_WINO4', '16')
    n, h, w, ci, co = shape
    rng = np.random.RandomState(ci * 7 + co)
    x = rng.randn(n, h, w, ci).astype(np.float32)
    wt = (rng.randn(3, 3, ci, co) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = (0.1 * rng.randn(co)).astype(np.float32)
    dy = rng.randn(n, h, w, co).astype(np.float32)
    y = _lib.op_conv2d_fwd(x, wt, b, True)
    assert np.array_equal(_lib.op_conv2d_fwd(x, wt, b, True), y)
    x64, w64, b64, dy64 = (t.astype(np.float64) for t in (x, wt, b, dy))
    ey = relerr(y, o.conv2d_fwd(x64, w64, b64, 'same'))
    print('short K loop', shape, 'y err %.2e' % ey)
    assert 4e-7 < ey < TOL_F4, 'outside the F(4x4,3x3) kernel\'s error band: %g' % ey
    dx, dw, db = _lib.op_conv2d_bwd(x, wt, dy, True)
    dx_ref, dw_ref, db_ref = o.conv2d_bwd(x64, w64, dy64, 'same')
    assert relerr(dx, dx_ref) < TOL_F4 and relerr(dw, dw_ref) < TOL and relerr(db, db_ref) < TOL
    dx2 = _lib.op_conv2d_bwd(x, wt, dy, True)[0]
    assert np.array_equal(dx2, dx)


@pytest.mark.parametrize('shape', [(3, 16, 18, 16, 64), (2, 17, 7, 32, 128), (5, 30, 21, 80, 64), (4, 16, 16, 48, 192),
                                   (7, 28, 28, 64, 64), (5, 16, 40, 32, 64), (3, 17, 70, 16, 128), (9, 18, 66, 32, 64)])
def test_winograd_bx6_short_loops_ragged_tiles_and_image_straddling(gpu_required, shape, monkeypatch):
    """conv_wino_bx6.hip at its edges: 1, 2, 3 and 5 sixteen-channel stages (the B pieces are single buffered behind counted
    vmcnt waits), ragged tile rows / columns, partial last tile blocks, and blocks of flat tile rows that straddle one or two
    image boundaries (14 and 8 tile rows per image under 16-row blocks) -- forward and data gradient against the float64
    oracle within the fp32 F(2x2,3x3) bound, twice (bit-identical)."""
    need_experiments()
    monkeypatch.setenv('L3_FP32_CONV', 'f2x2_bf16x6')
    n, h, w, ci, co = shape
    rng = np.random.RandomState(ci * 7 + co + h)
    x = rng.randn(n, h, w, ci).astype(np.float32)
    wt = (rng.randn(3, 3, ci, co) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = (0.1 * rng.randn(co)).astype(np.float32)
    dy = rng.randn(n, h, w, co).astype(np.float32)
    y = _lib.op_conv2d_fwd(x, wt, b, True)
    assert np.array_equal(_lib.op_conv2d_fwd(x, wt, b, True), y)
    x64, w64, b64, dy64 = (t.astype(np.float64) for t in (x, wt, b, dy))
    ey = relerr(y, o.conv2d_fwd(x64, w64, b64, 'same'))
    print('bx6 edge', shape, 'y err %.2e' % ey)
    assert ey < TOL
    if co % 16 == 0:
        dx, dw, db = _lib.op_conv2d_bwd(x, wt, dy, True)
        dx_ref, dw_ref, db_ref = o.conv2d_bwd(x64, w64, dy64, 'same')
        edx = relerr(dx, dx_ref)
        print('   dx err %.2e' % edx)
        assert edx < TOL and relerr(db, db_ref) < TOL
        assert np.array_equal(_lib.op_conv2d_bwd(x, wt, dy, True)[0], dx)


@pytest.mark.parametrize('case', [c for c in LEDGER_CONVS if c[3] >= 64], ids=[c[0] for c in LEDGER_CONVS if c[3] >= 64])
def test_weight_gradient_split_bf16_experiment(gpu_required, case, monkeypatch):
    """conv_wgrad_bx6.hip (debug knob L3_WG_BX6=1; NOT the product path -- 13.8 ms per step against 10.5 for the fp32 kernel,
    profiles/r05_bx6_ablations.txt): the F(3x3,2x2) weight gradient with both operands split into bfloat16 triples in registers,
    at the 14 real layer geometries, within the fp32 kernel's bound of the float64 oracle -- and it really is the other kernel
    (the results differ in the last bits)."""
    need_experiments()
    tag, h, w, ci, co = case
    x, wt, b, dy = _layer_data(*case)
    x64, w64, dy64 = (t.astype(np.float64) for t in (x, wt, dy))
    _, dw_ref, _ = o.conv2d_bwd(x64, w64, dy64, 'same', need_dx=False)
    dw32 = _lib.op_conv2d_bwd(x, wt, dy, True)[1]
    monkeypatch.setenv('L3_WG_BX6', '1')
    dw = _lib.op_conv2d_bwd(x, wt, dy, True)[1]
    e, e32 = relerr(dw, dw_ref), relerr(dw32, dw_ref)
    print(tag, 'dw split-bf16 %.2e  fp32 %.2e' % (e, e32))
    assert e < TOL and e < 2 * e32 + 1e-7
    assert not np.array_equal(dw, dw32)


@pytest.mark.parametrize('shape', [(16, 56, 56, 256, 256), (32, 28, 28, 512, 128), (8, 112, 112, 64, 64)])
def test_winograd_bx6_race_screen(gpu_required, shape, monkeypatch):
    """As test_winograd_f4_race_screen, for conv_wino_bx6.hip (LDS-DMA refills of the single-buffered filter pieces behind
    counted vmcnt waits, the A buffers behind one barrier per stage): sizes that keep every CU busy for several tile blocks,
    repeated, bit-identical -- and the first run within the fp32 F(2x2,3x3) bound of the float64 oracle on the first two and
    the last sample."""
    need_experiments()
    monkeypatch.setenv('L3_FP32_CONV', 'f2x2_bf16x6')
    n, h, w, ci, co = shape
    rng = np.random.RandomState(ci + h)
    x = np.maximum(rng.randn(n, h, w, ci), 0).astype(np.float32)
    wt = (rng.randn(3, 3, ci, co) * np.sqrt(2.0 / (9 * ci))).astype(np.float32)
    b = (0.1 * rng.randn(co)).astype(np.float32)
    y0 = _lib.op_conv2d_fwd(x, wt, b, True)
    for _ in range(6):
        assert np.array_equal(_lib.op_conv2d_fwd(x, wt, b, True), y0)
    ref = o.conv2d_fwd(x[:2].astype(np.float64), wt.astype(np.float64), b.astype(np.float64), 'same')
    assert relerr(y0[:2], ref) < TOL
    ref = o.conv2d_fwd(x[-1:].astype(np.float64), wt.astype(np.float64), b.astype(np.float64), 'same')
    assert relerr(y0[-1:], ref) < TOL


MP_CONVS = [c for c in LEDGER_CONVS if c[3] % 64 == 0]           # the 14 mixed-precision layers


@pytest.mark.parametrize('form', ['bf16', 'bf16_stored'])
@pytest.mark.parametrize('case', MP_CONVS, ids=[c[0] for c in MP_CONVS])
def test_conv_layer_bf16(gpu_required, case, form):
    """BASELINE configs[4] arithmetic, layer by layer: conv(bf16(x), bf16(w)) accumulated in fp32, forward,
    data gradient and weight gradient -- against oracle.mixed_precision('bf16') on the same inputs.  The
    products are exact in fp32, so only the summation order differs and the fp32 bound applies; and the
    result must be far from the unrounded fp32 convolution (it IS the rounded computation)."""
    tag, h, w, ci, co = case
    x, wt, b, dy = _layer_data(*case)
    x64, w64, b64, dy64 = (t.astype(np.float64) for t in (x, wt, b, dy))
    with o.mixed_precision('bf16'):
        y_ref = o.conv2d_fwd(x64, w64, b64, 'same')
        dx_ref, dw_ref, db_ref = o.conv2d_bwd(x64, w64, dy64, 'same')
    y = _lib.op_conv2d_fwd(x, wt, b, True, dtype=form)
    dx, dw, db = _lib.op_conv2d_bwd(x, wt, dy, True, dtype=form)
    errs = dict(y=relerr(y, y_ref), dx=relerr(dx, dx_ref), dw=relerr(dw, dw_ref), db=relerr(db, db_ref))
    print(tag, form, ' '.join('%s=%.2e' % kv for kv in errs.items()))
    assert max(errs.values()) < TOL, (tag, form, errs)
    assert relerr(y, o.conv2d_fwd(x64, w64, b64, 'same')) > 1e-4


BF16_OUT_CONVS = [c for c in LEDGER_CONVS if not c[0].endswith('4b')]     # every tower conv but the two embedding layers


@pytest.mark.parametrize('case', BF16_OUT_CONVS, ids=[c[0] for c in BF16_OUT_CONVS])
def test_conv_layer_bf16_stored_output(gpu_required, case):
    """Mixed-precision rule (2) of the oracle: a tower conv that feeds a BatchNormalization stores accumulator + bias
    as bfloat16 -- the mixed-precision layers and the first conv of each tower (fp32 arithmetic, conv_first.hip);
    the two embedding layers keep fp32.
    Every returned value must be a bfloat16, and it must be THE bfloat16 nearest to the float64 result except
    where the fp32 accumulation error (~1e-6 of the value, measured above) straddles a rounding boundary --
    then it is the neighbouring bfloat16.  Expected flip rate ~1e-6 / 2^-9 = a few 1e-4 of the elements."""
    tag, h, w, ci, co = case
    x, wt, b, dy = _layer_data(*case)
    x64, w64, b64 = (t.astype(np.float64) for t in (x, wt, b))
    with o.mixed_precision('bf16'):
        exact = o.conv2d_fwd(x64, w64, b64, 'same')
    want = o.bf16_round(exact)
    y = _lib.op_conv2d_fwd(x, wt, b, True, dtype='bf16_stored_out').astype(np.float64)
    assert np.array_equal(o.bf16_round(y), y)                                                # bfloat16 values
    # y = bf16(exact + e) with |e| <= TOL * max|exact| (the fp32 accumulation error bounded above), so it lies
    # within half a bfloat16 spacing of (exact + e): spacing at v = 2^(floor(log2|v|) - 7)
    tol = TOL * np.abs(exact).max()
    half_ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(exact), np.abs(y)) + 1e-300)) - 8)
    excess = np.abs(y - exact) - half_ulp
    frac = float((y != want).mean())
    print(tag, 'not the nearest bfloat16: %.2e of %d elements; worst |y - exact| - ulp/2 = %.2e of max (bound %.1e)'
          % (frac, want.size, float(excess.max() / np.abs(exact).max()), TOL))
    assert float(excess.max()) <= tol
    assert frac < 1e-3                     # near-ties only; measured 0.7e-4 .. 1.6e-4


@pytest.mark.parametrize('case', [c for c in MP_CONVS if c[3] == c[4] or c[0].endswith('a')][:8], ids=lambda c: c[0])
def test_conv_layer_bf16_stored_data_gradient(gpu_required, case):
    """Mixed-precision rule (3): the data gradient of a mixed-precision conv is stored as bfloat16.  Same criterion
    as for the stored forward output: bfloat16 values within half a spacing (+ the fp32 accumulation error) of
    the float64 result of the rounded-operand data gradient."""
    tag, h, w, ci, co = case
    x, wt, b, dy = _layer_data(*case)
    x64, w64, dy64 = (t.astype(np.float64) for t in (x, wt, dy))
    with o.mixed_precision('bf16'):
        exact, _, _ = o.conv2d_bwd(x64, w64, dy64, 'same')
    dx, dw, db = _lib.op_conv2d_bwd(x, wt, dy, True, dtype='bf16_stored_out')
    dx = dx.astype(np.float64)
    assert np.array_equal(o.bf16_round(dx), dx)
    tol = TOL * np.abs(exact).max()
    half_ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(exact), np.abs(dx)) + 1e-300)) - 8)
    excess = float((np.abs(dx - exact) - half_ulp).max())
    frac = float((dx != o.bf16_round(exact)).mean())
    print(tag, 'dx not the nearest bfloat16: %.2e of %d elements; worst excess %.2e of max' % (frac, dx.size, excess / np.abs(exact).max()))
    assert excess <= tol and frac < 1e-3


# (tag, H, W, C, pool padding, relu_mode) -- the Conv -> BN -> ReLU -> MaxPool(2,2) tails the engine fuses;
# relu_mode 2 is the Activation-before-BatchNormalization order of vision_model.py:137-139
POOL_TAILS = [('A.block1', 256, 199, 64, 0, 1), ('A.block2', 128, 99, 128, 0, 1), ('A.block3', 64, 49, 256, 0, 1),
              ('V.block1', 224, 224, 64, 1, 2), ('V.block2', 112, 112, 128, 1, 1), ('V.block3', 56, 56, 256, 1, 1)]


@pytest.mark.parametrize('xbf', [False, True], ids=['x_f32', 'x_bf16'])
@pytest.mark.parametrize('case', POOL_TAILS, ids=[c[0] for c in POOL_TAILS])
def test_bn_relu_pool_tail_layer(gpu_required, case, xbf):
    """x_bf16: the conv output arrives as bfloat16 storage (mixed-precision rule (2)); the kernels widen it on
    load, so on a bfloat16-valued x they must agree with the oracle exactly as the fp32-input kernels do."""
    tag, h, w, c, same, mode = case
    rng = np.random.RandomState(h + c + mode)
    x = (rng.randn(N, h, w, c) * 1.5 + 0.3).astype(np.float32)
    if xbf:
        x = o.bf16_round(x)
    g = (1 + 0.1 * rng.randn(c)).astype(np.float32)
    bt = (0.1 * rng.randn(c)).astype(np.float32)
    pad = 'same' if same else 'valid'
    x64, g64, b64 = x.astype(np.float64), g.astype(np.float64), bt.astype(np.float64)
    if mode == 1:
        y_ref, cache = o.bn_fwd(x64, g64, b64, None, None, True)
        r_ref = np.maximum(y_ref, 0)
        p_ref, pc = o.maxpool_fwd(r_ref, 2, 2, 2, 2, pad)
    else:
        r_in = np.maximum(x64, 0)
        y_ref, cache = o.bn_fwd(r_in, g64, b64, None, None, True)
        p_ref, pc = o.maxpool_fwd(y_ref, 2, 2, 2, 2, pad)
    p, mean, var = _lib.op_bn_relu_pool2_fwd(x, g, bt, same, relu_mode=mode, x_bf16=xbf)
    dp = (rng.randn(*p_ref.shape) * 1e-3).astype(np.float32)
    if xbf:
        dp = o.bf16_round(dp)                  # ... and so does the incoming gradient (rule (3))
    d_after_pool = o.maxpool_bwd(dp.astype(np.float64), pc)
    if mode == 1:
        dz = np.where(r_ref > 0, d_after_pool, 0)
        dx_ref, dg_ref, db_ref = o.bn_bwd(dz, g64, cache, True)
    else:
        dr, dg_ref, db_ref = o.bn_bwd(d_after_pool, g64, cache, True)
        dx_ref = np.where(x64 > 0, dr, 0)
    dx, dg, db, dbias = _lib.op_bn_relu_pool2_bwd(x, g, bt, dp, same, relu_mode=mode, x_bf16=3 if xbf else 0)
    errs = dict(p=relerr(p, p_ref), mean=relerr(mean, cache[2]), var=relerr(var, cache[3]), dx=relerr(dx, dx_ref),
                dgamma=relerr(dg, dg_ref), dbeta=relerr(db, db_ref))
    print(tag, ' '.join('%s=%.2e' % kv for kv in errs.items()))
    assert max(errs.values()) < TOL, (tag, errs)
    # the conv-bias gradient is the column sum of dx (analytically 0 in BN -> ReLU order)
    assert np.abs(dbias - dx_ref.reshape(-1, c).sum(0)).max() < 1e-4 * np.abs(dx_ref).sum(axis=(0, 1, 2)).max() + 1e-7


# (tag, rows-per-sample, C): Conv -> BN -> ReLU stages that are NOT followed by a 2x2 pool (first conv of each block)
BN_STAGES = [('A.bn1a', 256 * 199, 64), ('A.bn2a', 128 * 99, 128), ('A.bn3a', 64 * 49, 256), ('A.bn4a', 32 * 24, 512),
             ('V.bn1a', 224 * 224, 64), ('V.bn2a', 112 * 112, 128), ('V.bn3a', 56 * 56, 256), ('V.bn4a', 28 * 28, 512)]


@pytest.mark.parametrize('xbf', [False, True], ids=['x_f32', 'x_bf16'])
@pytest.mark.parametrize('case', BN_STAGES, ids=[c[0] for c in BN_STAGES])
def test_bn_relu_stage_layer(gpu_required, case, xbf):
    tag, rows, c = case
    rows *= N
    rng = np.random.RandomState(rows % 9973 + c)
    x = (rng.randn(rows, c) * 1.5 + 0.3).astype(np.float32)
    if xbf:
        x = o.bf16_round(x)
    g = (1 + 0.1 * rng.randn(c)).astype(np.float32)
    bt = (0.1 * rng.randn(c)).astype(np.float32)
    y_ref, cache = o.bn_fwd(x.astype(np.float64), g.astype(np.float64), bt.astype(np.float64), None, None, True)
    y_ref = np.maximum(y_ref, 0)
    y, mean, var = _lib.op_bn_relu_fwd(x, g, bt, 1, x_bf16=xbf)
    dy = (rng.randn(rows, c) * 1e-3).astype(np.float32)
    if xbf:
        dy = o.bf16_round(dy)
    dz = np.where(y > 0, dy, 0)                       # mask from the GPU's own y (borderline zeros)
    dx_ref, dg_ref, db_ref = o.bn_bwd(dz.astype(np.float64), g.astype(np.float64), cache, True)
    dx, dg, db = _lib.op_bn_relu_bwd(x, y, dy, g, mean, var, 1, beta=bt, x_bf16=3 if xbf else 0)       # beta given: the engine's fast kernels
    errs = dict(y=relerr(y, y_ref), mean=relerr(mean, cache[2]), var=relerr(var, cache[3]), dx=relerr(dx, dx_ref),
                dgamma=relerr(dg, dg_ref), dbeta=relerr(db, db_ref))
    print(tag, ' '.join('%s=%.2e' % kv for kv in errs.items()))
    assert max(errs.values()) < TOL, (tag, errs)
