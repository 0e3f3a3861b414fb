"""GPU parity tests: the HIP path (through the C ABI) against the numpy oracle and the
committed golden vectors.  Tolerances: logits within 1e-3 absolute (north-star bar, fp32);
elementwise ops to fp32 round-off; exact for integer->float preprocessing and max-pool."""
import ctypes
import importlib.util
import os
from collections import OrderedDict

import numpy as np
import pytest

from l3embedding_amd import _lib, model
from conftest import need_experiments
from oracle import l3_oracle as o

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
LOGIT_TOL = 1e-3
# Full-step gradient bounds against the float64 oracle (tests/golden).  Layer by layer the kernels agree with the
# oracle to 1e-5 (tests/test_layer_parity_gpu.py); through the whole network the distance is set by fp32
# conditioning (16 BatchNorm backward stages, max-pool arg-max decisions).  The bounds are ~3x the distances measured
# on MI355X (profiles/r03_parity_distances.txt); the golden files also hold the same distances for the float32
# NumPy oracle (`gd32:*`, written by make_golden.py), which is 6-140x further from float64 than the HIP path --
# and the test requires the HIP path to be at least that close.
# (max sampled |error| / tensor RMS, relative L2 error over the 256 samples, relative error of the tensor's L2 norm)
GRAD_BOUNDS = {
    'cnn_L3_melspec2_b2.npz': (0.15, 0.02, 0.02),      # measured 3.0e-2 / 3.5e-3 / 1.9e-3 (round 2: 5.7e-2 / 6.7e-3 / 6.7e-3; fp32 NumPy oracle: 0.74 / - / 0.20)
    'tiny_L3_b3.npz': (5e-5, 2e-5, 1e-5),              # measured 1.1e-5 / 4.7e-6 / 2.3e-6   (fp32 NumPy oracle: 9.6e-5 / - / 3.2e-5)
    'cnn_L3_orig_b1.npz': (0.25, 0.025, 3e-3),         # measured 7.9e-2 / 9.5e-3 / 1.0e-3   (fp32 NumPy oracle: 0.31 / - / 1.1e-2)
    # batch 8: BatchNorm statistics over 8 samples -- the well-conditioned case
    'cnn_L3_melspec2_b8.npz': (7e-2, 1.5e-2, 1.5e-2),  # measured 2.3e-2 / 4.9e-3 / 4.9e-3 (round 3, F(4x4,3x3) forward / data gradient)
    # batch 64: THE configuration the metric is quoted on (BASELINE.json configs[2], train.py:408-414 at train_batch_size = 64);
    # every BatchNorm normalises over 64 samples
    # every BatchNorm normalises over 64 samples.  The logits tighten no further (5e-5), and the gradient distances do not shrink with the
    # batch: they are set by fp32 sums over N*H*W elements (beta / gamma / bias gradients of the 28 x 28 layers: 50 176 terms that
    # cancel to a tenth of their RMS), which GROW with the batch; worst tensor vision_model/batch_normalization_7/beta
    'cnn_L3_melspec2_b64.npz': (0.15, 2.2e-2, 1e-2),   # measured 7.4e-2 / 1.0e-2 / 4.5e-3 (rounds 4, 5): 2x -- the distance is made of ReLU / arg-max
                                                       # flips and moves between 2e-2 and 8e-2 with the summation order (profiles/r05_parity_distances.txt)
    # the other two registry entries (model.py:220-262) at batch 2: a full training step each (round 4; before: forward only)
    'cnn_L3_kapredbinputbn_b2.npz': (0.15, 0.03, 0.03),  # measured 4.8e-2 / 9.2e-3 / 9.2e-3
    'cnn_L3_melspec1_b2.npz': (0.15, 0.08, 0.08),      # measured 4.0e-2 / 2.7e-2 / 2.7e-2 (the single-element beta of the audio input BatchNorm sets the last two)
}                                                      # batch 1: every BatchNorm normalises over a single sample's pixels
# three-step trajectory at batch 64 (lr 1e-5): relative loss distance per step measured 1.5e-7 / 5.5e-7 / 1.1e-5, logits 3.7e-5 / 6.0e-4 / 1.1e-3,
# inference logits after the three steps 2.4e-3 (logit scale 7.0); largest weight distance 2.00 Adam steps (one step taken the other way)
TRAJ_LOSS_TOL, TRAJ_LOGIT_TOL = 1e-4, 1e-2
# least fraction of the sampled entries of a step's gradient that the Adam step-1 comparison must cover (the entries whose sign the
# gradient bound leaves undetermined are masked out: that mask must not swallow the test -- ADVICE r03)
ADAM_MIN_COVER = 0.5
NP32_MEANINGFUL = ('cnn_L3_melspec2_b2.npz', 'tiny_L3_b3.npz', 'cnn_L3_orig_b1.npz')
# |w1 - w1_ref| / lr after the first Adam step, where |g| > 2 % of the tensor's largest sampled gradient
ADAM_STEP1 = {'cnn_L3_kapredbinputbn_b2.npz': 1e-3, 'cnn_L3_melspec1_b2.npz': 1e-3, 'cnn_L3_melspec2_b64.npz': 1e-3, 'cnn_L3_melspec2_b8.npz': 1e-3, 'cnn_L3_melspec2_b2.npz': 1e-3, 'tiny_L3_b3.npz': 1e-3, 'cnn_L3_orig_b1.npz': 5e-2}     # measured 1.5e-5, 1.5e-5, 1.1e-2


def _mod():
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLDEN, 'make_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# ---- operators ---------------------------------------------------------------------------------------
@pytest.mark.parametrize('shape', [
    (2, 9, 11, 16, 64, 3, 1), (1, 17, 13, 64, 128, 3, 1), (2, 12, 10, 1, 64, 3, 1), (2, 12, 10, 3, 64, 3, 1),
    (1, 8, 8, 128, 256, 3, 1), (2, 14, 9, 3, 10, 5, 0), (2, 14, 9, 10, 10, 5, 0), (3, 7, 5, 32, 48, 3, 1),
    (1, 6, 6, 256, 512, 3, 1), (1, 1, 300, 512, 70, 1, 0), (2, 5, 5, 64, 64, 3, 1), (1, 33, 31, 64, 64, 3, 1)])
def test_conv2d_fwd_bwd(gpu_required, shape):
    n, h, w, ci, co, k, same = shape
    rng = np.random.RandomState(sum(shape))
    x = rng.randn(n, h, w, ci).astype(np.float32)
    wt = (rng.randn(k, k, ci, co) / np.sqrt(k * k * ci)).astype(np.float32)
    b = rng.randn(co).astype(np.float32)
    pad = 'same' if same else 'valid'
    y_ref = o.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64), pad)
    y = _lib.op_conv2d_fwd(x, wt, b, same)
    f4 = k == 3 and same and ci >= 64 and ci % 8 == 0 and co % 64 == 0         # Winograd F(4x4,3x3): its own error budget
    assert relerr(y, y_ref) < (3e-5 if f4 else 5e-6)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dx_ref, dw_ref, db_ref = o.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64), pad)
    dx, dw, db = _lib.op_conv2d_bwd(x, wt, dy, same)
    f4d = k == 3 and same and co >= 64 and co % 8 == 0 and ci % 64 == 0        # the data gradient is a conv with Cin = co
    assert relerr(dx, dx_ref) < (3e-5 if f4d else 5e-6) and relerr(dw, dw_ref) < 5e-6 and relerr(db, db_ref) < 5e-6


@pytest.mark.parametrize('form', ['fma', 'x6'])
@pytest.mark.parametrize('shape', [(2, 12, 10, 1), (2, 12, 10, 3), (1, 5, 199, 1), (1, 7, 224, 3), (3, 4, 33, 3), (2, 3, 32, 1), (1, 40, 65, 3)])
def test_first_convolution_forward(gpu_required, shape, form, monkeypatch):
    """First convolution of a tower (audio_model.py:376-378, vision_model.py:130-132: 3x3 'same', 1 or 3 input channels -> 64;
    conv_first.hip) against the float64 oracle -- widths that are not multiples of the 32-pixel run, one-run rows, several images --
    and the impulse response tap by tap (row / column / channel order): the FMA kernel (27 / 9 v_fmac per output; the product path) and
    the split-bf16 matrix-core kernel of an L3_BUILD_EXPERIMENTS=1 library (x and w as three exact bfloat16 terms each, six products on
    v_mfma_f32_32x32x16_bf16: 20-25 % faster on the 3-channel layer, closer to float64, and not worth its downstream last-bit changes --
    profiles/r06_first_conv_mfma.txt, which also records the fp32-MFMA and packed-FMA forms that were slower)."""
    if form == 'x6':
        need_experiments()
        monkeypatch.setenv('L3_FIRST_FWD_X6', '1')
    n, h, w, ci = shape
    rng = np.random.RandomState(sum(shape))
    x = rng.randn(n, h, w, ci).astype(np.float32)
    wt = (rng.randn(3, 3, ci, 64) / np.sqrt(9 * ci)).astype(np.float32)
    b = rng.randn(64).astype(np.float32)
    y_ref = o.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64), 'same')
    y = _lib.op_conv2d_fwd(x, wt, b, True)
    assert relerr(y, y_ref) < 2e-6
    x1 = np.zeros((1, 5, 40, ci), np.float32)
    x1[0, 2, 33, ci - 1] = 1.0                                  # an impulse in the second 32-pixel run
    w1 = (np.arange(9 * ci * 64, dtype=np.float32).reshape(3, 3, ci, 64) + 1) / 512.0
    y1 = _lib.op_conv2d_fwd(x1, w1, np.zeros(64, np.float32), True)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            assert np.array_equal(y1[0, 2 + dy, 33 + dx], w1[1 - dy, 1 - dx, ci - 1]), (dy, dx)


def test_conv2d_empty_halo_and_identity(gpu_required):
    # A = I check with an asymmetric filter: catches row/col swaps in the MFMA output mapping
    x = np.zeros((1, 5, 5, 16), np.float32)
    x[0, 2, 3, 5] = 1.0
    w = np.arange(3 * 3 * 16 * 32, dtype=np.float32).reshape(3, 3, 16, 32) / 1000.0
    y = _lib.op_conv2d_fwd(x, w, np.zeros(32, np.float32), True)
    ref = o.conv2d_fwd(x.astype(np.float64), w.astype(np.float64), np.zeros(32), 'same')
    assert np.abs(y - ref).max() < 1e-6
    assert y[0, 1, 2, 7] == pytest.approx(w[2, 2, 5, 7])     # output (1,2) sees the impulse through tap (2,2)


@pytest.mark.parametrize('rows,c,relu', [(1000, 64, 1), (333, 1, 0), (777, 3, 0), (4096, 128, 0), (100, 512, 1), (50, 10, 1), (5000, 256, 1)])
def test_batchnorm_fwd_bwd(gpu_required, rows, c, relu):
    rng = np.random.RandomState(rows + c)
    x = (rng.randn(rows, c) * 3 + 5).astype(np.float32)
    g = (rng.rand(c) + 0.5).astype(np.float32)
    bt = rng.randn(c).astype(np.float32)
    y_ref, cache = o.bn_fwd(x.astype(np.float64), g.astype(np.float64), bt.astype(np.float64), None, None, True)
    if relu:
        y_ref = np.maximum(y_ref, 0)
    y, mean, var = _lib.op_bn_relu_fwd(x, g, bt, relu)
    assert relerr(y, y_ref) < 2e-6 and relerr(mean, cache[2]) < 1e-6 and relerr(var, cache[3]) < 5e-6
    dy = rng.randn(rows, c).astype(np.float32)
    dz = np.where(y > 0, dy, 0) if relu else dy           # mask from the GPU's own y (borderline zeros)
    dx_ref, dg_ref, db_ref = o.bn_bwd(dz.astype(np.float64), g.astype(np.float64), cache, True)
    dx, dg, db = _lib.op_bn_relu_bwd(x, y, dy, g, mean, var, relu)
    assert relerr(dx, dx_ref) < 1e-5 and relerr(dg, dg_ref) < 1e-5 and relerr(db, db_ref) < 1e-5


@pytest.mark.parametrize('nblk,c', [(3841, 64), (3841, 512), (2049, 8), (4097, 128), (25088, 64), (2304, 256), (700, 64)])
def test_batchnorm_moments_from_epilogue_partials(gpu_required, nblk, c):
    """Second reduction stage of the conv-epilogue BatchNorm partials (bn_fused.hip launch_fast_final).  Above 2048 rows a
    pre-reduction leaves fp64 sums IN PLACE, two float rows wide: a chunk of ONE row (nblk % rows-per-chunk == 1, e.g. 3841)
    used to write one row past the buffer.  The operator runs on a buffer of exactly nblk rows and fails if the
    guard behind it changed; results against float64 sums."""
    rng = np.random.RandomState(nblk + c)
    rows_per = 256
    part = np.empty((nblk, 2, c), np.float32)
    part[:, 0] = rng.randn(nblk, c) * 16 + 3
    part[:, 1] = rng.rand(nblk, c) * 4000 + 300
    pivot = rng.randn(c).astype(np.float32)
    rows = nblk * rows_per
    mean, var = _lib.op_bn_stats_from_partials(part, pivot, rows)
    s0, s1 = part[:, 0].astype(np.float64).sum(0), part[:, 1].astype(np.float64).sum(0)
    dm = s0 / rows
    assert relerr(mean, pivot.astype(np.float64) + dm) < 1e-6
    assert relerr(var, np.maximum(s1 / rows - dm * dm, 0)) < 1e-6


@pytest.mark.parametrize('cfg', [(2, 8, 8, 64, 2, 2, 0), (2, 9, 7, 16, 2, 2, 0), (2, 9, 7, 16, 2, 2, 1), (2, 32, 24, 8, 32, 24, 0),
                                 (1, 28, 28, 4, 28, 28, 1), (2, 32, 24, 4, 8, 8, 1), (2, 10, 11, 10, 3, 3, 0), (1, 28, 28, 4, 7, 7, 1),
                                 (1, 199, 5, 3, 2, 2, 0), (3, 32, 24, 128, 32, 24, 0), (2, 28, 28, 64, 28, 28, 0),
                                 (2, 28, 28, 512, 28, 28, 1)])
def test_maxpool_exact(gpu_required, cfg):
    n, h, w, c, ph, pw, same = cfg
    rng = np.random.RandomState(sum(cfg))
    x = rng.randn(n, h, w, c).astype(np.float32)
    x[0, :2, :2, 0] = 1.5          # a tie: first element in scan order must win
    if c >= 64 and h >= 28:        # global pools see post-ReLU maps: whole channels of exact zeros and ties
        x = np.maximum(x, 0)
        x[:, :, :, 5] = 0.0
        x[0, 3, 7, 9] = x[0, 20, 1, 9] = 9.0
    y_ref, cache = o.maxpool_fwd(x, ph, pw, ph, pw, 'same' if same else 'valid')
    y = _lib.op_maxpool_fwd(x, ph, pw, ph, pw, same)
    assert np.array_equal(y, y_ref)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    assert np.array_equal(_lib.op_maxpool_bwd(x, dy, ph, pw, ph, pw, same), o.maxpool_bwd(dy, cache))


@pytest.mark.parametrize('cfg', [(2, 8, 8, 64, 0), (2, 9, 7, 16, 0), (2, 9, 7, 16, 1), (1, 199, 6, 64, 0), (3, 10, 11, 128, 1), (1, 4, 4, 512, 1)])
def test_fused_bn_relu_pool2(gpu_required, cfg):
    """Conv-BN-ReLU-MaxPool tail as one kernel (full-resolution activation never stored):
    forward and backward against the unfused oracle ops, odd sizes, 'valid' and 'same'."""
    n, h, w, c, same = cfg
    rng = np.random.RandomState(sum(cfg))
    x = (rng.randn(n, h, w, c) * 2 + 0.5).astype(np.float32)
    x[0, :2, :2, 0] = 3.0                                    # a tie inside one window
    g = (rng.rand(c) + 0.5).astype(np.float32)
    bt = (rng.randn(c) * 0.3).astype(np.float32)
    pad = 'same' if same else 'valid'
    y_ref, cache = o.bn_fwd(x.astype(np.float64), g.astype(np.float64), bt.astype(np.float64), None, None, True)
    r_ref = np.maximum(y_ref, 0)
    p_ref, pc = o.maxpool_fwd(r_ref, 2, 2, 2, 2, pad)
    p, mean, var = _lib.op_bn_relu_pool2_fwd(x, g, bt, same)
    assert p.shape == p_ref.shape and relerr(p, p_ref) < 2e-6
    dp = rng.randn(*p_ref.shape).astype(np.float32)
    dr = o.maxpool_bwd(dp.astype(np.float64), pc)
    dz = np.where(r_ref > 0, dr, 0)
    dx_ref, dg_ref, db_ref = o.bn_bwd(dz, g.astype(np.float64), cache, True)
    dx, dg, db, dbias = _lib.op_bn_relu_pool2_bwd(x, g, bt, dp, same)
    assert relerr(dx, dx_ref) < 2e-5 and relerr(dg, dg_ref) < 2e-5 and relerr(db, db_ref) < 2e-5
    assert np.abs(dbias).max() < 1e-3 * np.abs(dx_ref).sum(axis=(0, 1, 2)).max() + 1e-4     # sum of dx: analytically 0
    assert np.abs(dbias - dx.reshape(-1, c).sum(0)).max() < 1e-3


def test_preprocess_bit_exact(gpu_required):
    u8 = np.arange(256, dtype=np.uint8)
    i16 = np.concatenate([np.array([-32768, -1, 0, 1, 32767], np.int16),
                          np.random.RandomState(0).randint(-32768, 32768, 4096).astype(np.int16)])
    vo, ao = _lib.op_preprocess(u8, i16)
    assert np.array_equal(vo, o.preprocess_video(u8))
    assert np.array_equal(ao, o.pcm2float(i16, np.float32))
    assert vo[0] == -1.0 and vo[255] == 1.0 and ao[0] == -1.0 and ao[2] == 0.0


@pytest.mark.parametrize('mt', ['cnn_L3_melspec2', 'cnn_L3_melspec1', 'cnn_L3_orig', 'cnn_L3_kapredbinputbn', 'tiny_L3'])
def test_frontend(gpu_required, mt):
    v, a, l = o.synthetic_batch(2, seed=9)
    t = np.arange(48000) / 48000.0
    a[1, 0] = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.05 * np.sin(2 * np.pi * 5000 * t)).astype(np.float32)
    kind = o.model_spec(mt)['frontend']
    ref = o.frontend_forward(kind, a, None, 'sample', np.float64)
    got = _lib.op_frontend(mt, a)
    assert got.shape == ref.shape
    # noise-like input: everything is far above the fp32 round-off floor of the 2048-tap DFT
    assert np.abs(got[0] - ref[0]).max() < 5e-3
    if o.FRONTENDS[kind]['db']:
        # tonal input: compare relative amplitude (dB values below ~-55 are fp32 round-off in any implementation)
        assert np.abs(10 ** (got[1] / 10) - 10 ** (ref[1] / 10)).max() < 2e-5
        assert got.max() == 0.0 and got.min() >= -80.0
    else:
        hi = ref[1] > ref[1].max() - 1.5
        assert np.abs(got[1][hi] - ref[1][hi]).max() < 1e-3


@pytest.mark.parametrize('mt', ['cnn_L3_melspec2', 'cnn_L3_melspec1'])
def test_factored_dft_equals_the_dft_gemm_and_yields_to_edited_kernels(gpu_required, mt, monkeypatch):
    """kapre's Melspectrogram (audio_model.py:257-260,367-369) is a strided convolution with Hann-windowed cos / -sin kernels that
    kapre does not train.  When the engine holds those STOCK kernels (bit for bit what MODELS[...]() generates and every reference
    weight file stores) the 2048-point transform runs factored, 2048 = 32 x 64, as two small GEMMs around a twiddle pass
    (frontend.hip dft_*); otherwise -- a weight file with edited kernels -- as the GEMM against the kernels as given.
    (1) factored against the full GEMM (L3_DFT_FACTORED=0) on noise and on tones: the same power spectrum to fp32 round-off, and
    both within the front-end's bound of the float64 oracle; (2) an engine whose real_kernels were edited must follow the EDITED
    kernels (the oracle with the same constants), i.e. the factored path must have stepped aside."""
    v, a, l = o.synthetic_batch(3, seed=9)
    t = np.arange(48000) / 48000.0
    a[1, 0] = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.05 * np.sin(2 * np.pi * 5000 * t)).astype(np.float32)
    a[2, 0] = np.where(np.arange(48000) % 2 == 0, 0.3, -0.3).astype(np.float32) * (1 + 0.1 * np.sin(2 * np.pi * 3 * t)).astype(np.float32)   # energy at the Nyquist bin
    kind = o.model_spec(mt)['frontend']
    ref = o.frontend_forward(kind, a, None, 'sample', np.float64)
    got = {}
    for mode in ('1', '2', '0'):         # one fused kernel (the product form) | the two-GEMM form of the same factorisation | the full GEMM
        monkeypatch.setenv('L3_DFT_FACTORED', mode)
        got[mode] = _lib.op_frontend(mt, a)
    lin = lambda x: 10 ** (x / 10)
    for mode in ('1', '2', '0'):
        assert np.abs(got[mode][0] - ref[0]).max() < 5e-3, mode                       # noise: dB values
        assert np.abs(lin(got[mode][1:]) - lin(ref[1:])).max() < 2e-5, mode           # tones: relative amplitude
    for mode in ('1', '2'):
        d_noise = float(np.abs(got[mode][0] - got['0'][0]).max())
        d_tone = float(np.abs(lin(got[mode][1:]) - lin(got['0'][1:])).max())
        print('%s: factored (%s) vs GEMM DFT: noise %.2e dB, tones %.2e of the peak amplitude; vs float64: %.2e / %.2e dB'
              % (mt, 'one kernel' if mode == '1' else 'two GEMMs', d_noise, d_tone, np.abs(got[mode][0] - ref[0]).max(), np.abs(got['0'][0] - ref[0]).max()))
        assert d_noise < 2e-3 and d_tone < 1e-5
        assert not np.array_equal(got[mode], got['0'])         # it really is the other path
    # (2) edited kernels: scale the real kernels of the bins 100..199 by 2 -- their power grows, the factored path may not be used
    monkeypatch.delenv('L3_DFT_FACTORED')
    consts = o.frontend_constants(kind)
    consts['real_kernels'] = consts['real_kernels'].copy()
    consts['real_kernels'][..., 100:200] *= 2.0
    ref2 = o.frontend_forward(kind, a[:1], consts, 'sample', np.float64)
    eng = _lib.Engine(mt, 1, seed=0)
    name = [n for n, _, _ in eng.param_table() if n.endswith('/real_kernels')][0]
    eng.set_param(name, consts['real_kernels'])
    eng.forward(v[:1], a[:1], training=False)
    got2 = eng.activation('audio_model/frontend').reshape(ref2.shape)
    eng.close()
    assert np.abs(got2 - ref2).max() < 5e-3
    assert np.abs(ref2 - ref[:1]).max() > 1.0                  # the edit is visible in the output at all


def test_frontend_batch_scope(gpu_required):
    v, a, l = o.synthetic_batch(2, seed=10)
    a[1] *= 0.01
    ref = o.frontend_forward('melspec2', a, None, 'batch', np.float64)
    got = _lib.op_frontend('cnn_L3_melspec2', a, db_max_scope='batch')
    assert np.abs(got - ref).max() < 5e-3 and got[1].max() < -15


# ---- full model against the golden vectors -----------------------------------------------------------------
def _engine_from_golden(fname, **kw):
    z = np.load(os.path.join(GOLDEN, fname))
    mod = _mod()
    mt, B = str(z['model_type']), int(z['batch'])
    P = mod.perturbed_params(mt, int(z['param_seed']))
    v, a, l = o.synthetic_batch(B, seed=int(z['data_seed']))
    eng = _lib.Engine(mt, B, **kw)
    assert [n for n, _, _ in eng.param_table()] == [n for n, _, _, _ in o.param_table(mt)]
    for n, s, _ in eng.param_table():     # constants generated in C++ must equal kapre's / librosa's
        if '/real_kernels' in n or '/imag_kernels' in n or '/freq2mel' in n:
            assert np.abs(eng.get_param(n, s) - P[n]).max() < 1e-7, n
    eng.set_params(P)
    return z, mod, mt, B, P, (v, a, l), eng


@pytest.mark.parametrize('fname', ['cnn_L3_melspec2_b2.npz', 'tiny_L3_b3.npz', 'cnn_L3_orig_b1.npz', 'cnn_L3_melspec2_b8.npz', 'cnn_L3_melspec2_b64.npz',
                                   'cnn_L3_kapredbinputbn_b2.npz', 'cnn_L3_melspec1_b2.npz'])
def test_training_step_matches_golden(gpu_required, fname):
    z, mod, mt, B, P, (v, a, l), eng = _engine_from_golden(fname)
    probs, logits = eng.forward(v, a, training=False)
    d_eval = float(np.abs(logits - z['eval_logits']).max())
    assert d_eval < LOGIT_TOL
    assert np.abs(probs - z['eval_probs']).max() < LOGIT_TOL
    probs, logits = eng.forward(v, a, training=True)
    d_train = float(np.abs(logits - z['train_logits']).max())
    print('%s: |logits - float64 oracle| inference %.2e, training %.2e (bar %.0e; logit scale %.2f)' % (
        fname, d_eval, d_train, LOGIT_TOL, float(np.abs(z['train_logits']).max())))
    assert d_train < LOGIT_TOL
    # eval step: same loss definition with inference-mode BN
    ev = o.forward(mt, P, v, a, False, np.float64)
    q = np.clip(ev['probs'], 1e-7, 1 - 1e-7)
    ev_loss = float((-(l * np.log(q)).sum(1)).mean() + o.l2_penalty(P, mt))
    loss_e, acc_e = eng.eval_step(v, a, l)
    assert abs(loss_e - ev_loss) < 2e-3 * max(1, abs(ev_loss))
    # one training step
    loss, acc = eng.train_step(v, a, l, float(z['lr']))
    assert abs(loss - float(z['loss'])) < 1e-3 * max(1.0, abs(float(z['loss'])))
    assert acc == pytest.approx(float(z['acc']))
    G = eng.get_grads()
    W1 = eng.get_params()
    bad = []
    worst = dict(err=0.0, l2=0.0, nerr=0.0, w1=0.0)
    covered = total = 0
    for n, _, tr in eng.param_table():
        if not tr:
            continue
        gnorm = float(z['gnorm:' + n])
        idx = mod.sample_idx(n, G[n].size)
        if gnorm < 1e-7:
            continue          # conv biases feeding a BatchNorm: analytically zero gradient
        ref = z['gsamp:' + n]
        got = G[n].astype(np.float64)
        err, nerr = mod.grad_metrics(got, ref, gnorm, idx)              # max sampled error / RMS, norm error
        l2 = float(np.sqrt(((got.ravel()[idx] - ref) ** 2).sum() / ((ref ** 2).sum() + 1e-300)))
        # A single-element tensor (gamma / beta of the one-channel audio input BatchNorm) is ONE sum over every pixel of the batch: where
        # its terms cancel, no fp32 order of summation lands close in RELATIVE terms -- the float32 NumPy restatement itself is more
        # than its own size away (gd32 > 1; cnn_L3_melspec1 batch 2: 2.3).  Such an entry must be at least twice as close as that
        # restatement instead, and stays out of the file's worst-case figures.
        np32_err = float(z['gd32:' + n][0]) if 'gd32:' + n in z.files else 0.0
        if got.size == 1 and np32_err > 1.0:
            print('%s: %s is ill-conditioned (float32 NumPy %.2f of its size away): HIP path %.2f' % (fname, n, np32_err, err))
            if err > 0.5 * np32_err:
                bad.append((n, err, 'ill-conditioned scalar', np32_err))
            continue
        worst['err'], worst['l2'], worst['nerr'] = max(worst['err'], err), max(worst['l2'], l2), max(worst['nerr'], nerr)
        if err > GRAD_BOUNDS[fname][0] or l2 > GRAD_BOUNDS[fname][1] or nerr > GRAD_BOUNDS[fname][2]:
            bad.append((n, err, l2, nerr))
        # Adam step 1 moves a weight by lr*g/(|g| + eps'), eps' = 1e-8/sqrt(1-beta2): ~lr*sign(g), so compare where
        # the sign is determined (|g| above 2 % of the tensor's largest sampled gradient
        # ... and above twice the gradient distance this file's bound allows -- an entry inside it may change sign)
        rms = gnorm / np.sqrt(got.size)
        ok = np.abs(ref) > max(0.02 * np.abs(ref).max(), 2 * GRAD_BOUNDS[fname][0] * rms)
        covered, total = covered + int(ok.sum()), total + ok.size
        if ok.any():
            w1 = float(np.abs(W1[n].ravel()[idx][ok] - z['w1samp:' + n][ok]).max()) / float(z['lr'])
            worst['w1'] = max(worst['w1'], w1)
            if w1 > ADAM_STEP1[fname]:
                bad.append((n, 'adam step', w1))
    print('%s: worst grad err/rms %.2e, sampled L2 %.2e, norm %.2e; Adam step-1 error %.2e lr  (fp32 NumPy: %.2e / - / %.2e)' % (
        fname, worst['err'], worst['l2'], worst['nerr'], worst['w1'],
        max(float(z[k][0]) for k in z.files if k.startswith('gd32:')),
        max(float(z[k][1]) for k in z.files if k.startswith('gd32:'))))
    print('%s: Adam step-1 comparison covers %d of %d sampled entries (%.0f %%)' % (fname, covered, total, 100.0 * covered / max(1, total)))
    # absolute bounds (GRAD_BOUNDS, ADAM_STEP1: ~3x the measured distances) -- the float32 NumPy restatement's own distance is printed
    # for scale only: at batch 8 and 64 it is O(1) of the RMS (its BatchNorm moments are naive fp32 sums over 10^5..10^6 elements), so
    # "at least as close as fp32 NumPy" could never fail there
    assert not bad, bad
    assert covered >= ADAM_MIN_COVER * total, (covered, total)
    if fname in NP32_MEANINGFUL:
        # at batch 1-3 the float32 NumPy restatement is a meaningful yardstick (its BatchNorm moments are short sums): the HIP path
        # must be at least as close to float64 as that restatement is
        np32_worst = max(float(z[k][0]) for k in z.files if k.startswith('gd32:') and z[k].size and float(z['gnorm:' + k[5:]]) >= 1e-7)
        assert worst['err'] <= np32_worst, (worst['err'], np32_worst)
    for n, s, tr in eng.param_table():
        if n.endswith('/moving_mean') or n.endswith('/moving_variance'):
            ref = z['mov:' + n]
            assert np.abs(W1[n] - ref).max() < 2e-3 * max(1.0, np.abs(ref).max()), n
    eng.close()


def test_split_bf16_engine_matches_the_golden_step(gpu_required):
    """l3_config.fp32_conv = L3_FP32_CONV_F2X2_BF16X6 through a whole training step at batch 8 (forward and data gradient of the 14
    layers on split-bf16 operands, conv_wino_bx6.hip): logits, loss and every sampled gradient within the SAME bounds the fp32
    engines are held to -- it is an fp32-grade configuration, not a reduced-precision one."""
    need_experiments()
    fname = 'cnn_L3_melspec2_b8.npz'
    z, mod, mt, B, P, (v, a, l), eng = _engine_from_golden(fname, fp32_conv='f2x2_bf16x6')
    _, logits = eng.forward(v, a, training=True)
    d_train = float(np.abs(logits - z['train_logits']).max())
    _, logits_e = eng.forward(v, a, training=False)
    d_eval = float(np.abs(logits_e - z['eval_logits']).max())
    loss, acc = eng.train_step(v, a, l, float(z['lr']))
    G = eng.get_grads()
    worst = 0.0
    for n, _, tr in eng.param_table():
        if not tr or float(z['gnorm:' + n]) < 1e-7 or G[n].size == 1:
            continue
        idx = mod.sample_idx(n, G[n].size)
        err, nerr = mod.grad_metrics(G[n].astype(np.float64), z['gsamp:' + n], float(z['gnorm:' + n]), idx)
        worst = max(worst, err)
        assert err <= GRAD_BOUNDS[fname][0] and nerr <= GRAD_BOUNDS[fname][2], (n, err, nerr)
    print('split-bf16 engine, %s: |logits - float64| training %.2e inference %.2e, loss %.6f (oracle %.6f), worst grad err/rms %.2e' % (
        fname, d_train, d_eval, loss, float(z['loss']), worst))
    assert d_train < LOGIT_TOL and d_eval < LOGIT_TOL
    assert abs(loss - float(z['loss'])) < 1e-3 * max(1.0, abs(float(z['loss'])))
    eng.close()


def test_three_training_steps_at_batch_64_match_the_float64_trajectory(gpu_required):
    """BASELINE configs[2] over time: three consecutive fit_generator steps (l3embedding/train.py:408-414 at train_batch_size = 64) --
    Adam moments, BatchNorm moving statistics and zero-debias accumulators carried across steps -- against the float64 oracle's
    trajectory (tests/golden/make_traj_golden.py), then an inference-mode forward of a fourth batch through the moving statistics.
    Step 1 must agree like the single-step golden; after it, every Adam step moves every weight by ~lr * sign(g), so entries whose
    gradient sign is inside the fp32 noise take the other branch and the trajectories separate slowly: the bounds below are ~3x the
    measured distances (profiles/r04_parity_distances.txt)."""
    z = np.load(os.path.join(GOLDEN, 'cnn_L3_melspec2_b64_traj.npz'))
    mod = _mod()
    mt, B, lr, steps = str(z['model_type']), int(z['batch']), float(z['lr']), int(z['steps'])
    P = mod.perturbed_params(mt, int(z['param_seed']))
    eng = _lib.Engine(mt, B)
    eng.set_params(P)
    dl, dlog = [], []
    for s_ in range(steps):
        v, a, l = o.synthetic_batch(B, seed=int(z['data_seed']) + s_)
        _, logits = eng.forward(v, a, training=True)
        dlog.append(float(np.abs(logits - z['logits%d' % s_]).max()))
        loss, acc = eng.train_step(v, a, l, lr)
        dl.append(abs(loss - float(z['loss'][s_])) / max(1.0, abs(float(z['loss'][s_]))))
    v, a, l = o.synthetic_batch(B, seed=int(z['data_seed']) + steps)
    _, logits = eng.forward(v, a, training=False)
    d_eval = float(np.abs(logits - z['eval_logits']).max())
    W = eng.get_params()
    dw = 0.0
    for n, _, tr in eng.param_table():
        idx = mod.sample_idx(n, W[n].size)
        ref = z['w:' + n]
        scale = lr if tr else max(1e-6, float(np.abs(ref).max()))          # trainable: in units of one Adam step; moving statistics: relative
        dw = max(dw, float(np.abs(W[n].ravel()[idx] - ref).max()) / scale if tr else 0.0)
        if not tr and ('moving' in n):
            assert np.abs(W[n].ravel()[idx] - ref).max() < 2e-3 * max(1.0, np.abs(ref).max()), n
    print('batch-64 trajectory: |logits - float64| per step %s, loss distance per step %s, inference logits after %d steps %.2e (scale %.2f), '
          'largest weight distance %.2f Adam steps' % (['%.1e' % x for x in dlog], ['%.1e' % x for x in dl], steps, d_eval,
                                                      float(np.abs(z['eval_logits']).max()), dw))
    eng.close()
    assert dlog[0] < LOGIT_TOL and dl[0] < 1e-4
    assert max(dl) < TRAJ_LOSS_TOL and max(dlog) < TRAJ_LOGIT_TOL and d_eval < TRAJ_LOGIT_TOL
    assert dw <= 2.0 * steps + 0.5          # no weight further from the oracle's than every step taken the other way


@pytest.mark.parametrize('mt', ['cnn_L3_kapredbinputbn', 'cnn_L3_melspec1'])
def test_other_registry_models_forward(gpu_required, mt):
    mod = _mod()
    P = mod.perturbed_params(mt, 31)
    v, a, l = o.synthetic_batch(1, seed=32)
    eng = _lib.Engine(mt, 1)
    eng.set_params(P)
    for training in (False, True):
        ref = o.forward(mt, P, v, a, training, np.float64)
        probs, logits = eng.forward(v, a, training=training)
        assert np.abs(logits - ref['logits']).max() < LOGIT_TOL
    eng.close()


def test_multi_step_training_trajectory(gpu_required):
    mt, B, lr = 'tiny_L3', 4, 1e-3
    mod = _mod()
    P = mod.perturbed_params(mt, 41)
    eng = _lib.Engine(mt, B)
    eng.set_params(P)
    adam, bn = o.AdamState(), o.BNMovingState(True)
    for step in range(4):
        v, a, l = o.synthetic_batch(B, seed=50 + step)
        ref = o.train_step(mt, P, adam, bn, v, a, l, lr, np.float64)
        loss, acc = eng.train_step(v, a, l, lr)
        assert abs(loss - ref['loss']) < 5e-3 * max(1.0, abs(ref['loss'])), (step, loss, ref['loss'])
    # weights still agree after 4 Adam steps (skip biases that feed a BN: noise-driven in fp32)
    W = eng.get_params()
    spec = o.model_spec(mt)
    for n in W:
        if n.endswith('/kernel') or n.endswith('/gamma') or n.endswith('/beta') or n.startswith('dense'):
            assert np.abs(W[n] - P[n]).max() < 3 * lr, n
    v, a, l = o.synthetic_batch(B, seed=60)
    ref = o.forward(mt, P, v, a, False, np.float64)
    probs, logits = eng.forward(v, a, training=False)
    assert np.abs(logits - ref["logits"]).max() < 1e-1      # 4 sign-like Adam steps amplify fp32 gradient noise
    eng.close()


def test_embeddings_match_oracle(gpu_required):
    mt = 'cnn_L3_melspec2'
    mod = _mod()
    P = mod.perturbed_params(mt, 51)
    v, a, l = o.synthetic_batch(3, seed=52)
    eng = _lib.Engine(mt, 2)            # 3 frames through a batch-2 engine: exercises chunking
    eng.set_params(P)
    for pooling, dim in (('original', 6144), ('short', 512)):
        ref = o.embed_audio(mt, P, a, pooling, np.float64)
        got = eng.embed_audio(a, o.AUDIO_POOLING[mt][pooling])
        assert got.shape == (3, dim)
        assert np.abs(got - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())
    refv = o.embed_vision(mt, P, v, np.float64)
    gotv = eng.embed_vision(v)
    assert gotv.shape == (3, 8192) and np.abs(gotv - refv).max() < 2e-3 * max(1.0, np.abs(refv).max())
    eng.close()


def test_reference_entry_points_roundtrip(gpu_required, tmp_path):
    """MODELS -> compile -> train_on_batch -> save_weights -> load_model/load_embedding -> predict."""
    mt = 'cnn_L3_melspec2'
    m, inputs, out = model.MODELS[mt]()
    m.compile(model.Adam(lr=1e-4), loss='categorical_crossentropy', metrics=['accuracy'])
    v, a, l = o.synthetic_batch(2, seed=61)
    loss, acc = m.train_on_batch([v, a], l)
    assert np.isfinite(loss)
    path = str(tmp_path / 'model_latest.h5')
    m.save_weights(path)
    m2 = model.load_model(path, mt)
    w1, w2 = m.get_weights(), m2.get_weights()
    assert len(w1) == 111 and all(np.array_equal(x, y) for x, y in zip(w1, w2))
    p1, p2 = m.predict([v, a]), m2.predict([v, a])
    assert np.array_equal(p1, p2) and np.allclose(p1.sum(1), 1, atol=1e-6)
    emb = model.load_embedding(path, mt, 'audio', 'original')
    e = emb.predict(a)
    assert e.shape == (2, 6144)
    with pytest.raises(ValueError, match='Invalid embedding type'):
        model.load_embedding(path, mt, 'smell', 'original')


def test_train_entry_point_and_resume(gpu_required, tmp_path):
    """train(...) as 03_train_embedding.py calls it: run directory <out>/embedding/<subset>/<model_type>/<ts>
    (train.py:231-234,277) with its artefacts, checkpoints in the keras HDF5 layout, then
    --continue-model-dir resume (train.py:218-421).  train and validation batch sizes differ on purpose."""
    from l3embedding_amd import h5lite, train as T
    rng = np.random.RandomState(5)
    for split in ('tiny_train', 'tiny_valid'):
        d = tmp_path / 'data' / split
        d.mkdir(parents=True)
        for i in range(2):
            root = h5lite.Group()
            lab = rng.randint(0, 2, 6)
            root.create_dataset('audio', rng.randint(-32768, 32768, (6, 1, 48000)).astype(np.int16), compression='gzip')
            root.create_dataset('video', rng.randint(0, 256, (6, 224, 224, 3)).astype(np.uint8), compression='gzip')
            root.create_dataset('label', np.stack([lab, 1 - lab], 1).astype(np.int64), compression='gzip')
            h5lite.write_file(str(d / ('%d_%d_0.h5' % (20171021 + i, i))), root)
    out = str(tmp_path / 'out')
    tr_dir, va_dir = str(tmp_path / 'data' / 'tiny_train'), str(tmp_path / 'data' / 'tiny_valid')
    T.train(tr_dir, va_dir, out, num_epochs=2, train_epoch_size=2, validation_epoch_size=1, train_batch_size=4,
            validation_batch_size=3, model_type='tiny_L3', learning_rate=1e-3, checkpoint_interval=1, gpus=1)
    runs = os.listdir(os.path.join(out, 'embedding', 'tiny', 'tiny_L3'))
    assert len(runs) == 1
    md = os.path.join(out, 'embedding', 'tiny', 'tiny_L3', runs[0])
    assert T.embedding_desc_str(md).split('/')[-1] == 'tiny_L3'            # 05_generate_embedding_samples.py:144-153
    for f in ('config.json', 'model.json', 'model_spec.pkl', 'model_latest.h5', 'model_best_valid_accuracy.h5',
              'model_best_valid_loss.h5', 'model_checkpoint.01.h5', 'model_checkpoint.02.h5', 'history_csvlog.csv',
              'history_checkpoint.pkl', 'history.pkl'):
        assert os.path.exists(os.path.join(md, f)), f
    import json
    cfg = json.load(open(os.path.join(md, 'config.json')))
    assert cfg['model_id'] == 'tiny/tiny_L3' and cfg['model_dir'] == md and cfg['train_batch_size'] == 4
    assert T.get_restart_info(os.path.join(md, 'history_csvlog.csv'))[0] == 1
    m = model.load_model(os.path.join(md, 'model_latest.h5'), 'tiny_L3')
    assert len(m.get_weights()) == 42
    T.train(tr_dir, va_dir, out, num_epochs=3, train_epoch_size=2, validation_epoch_size=1, train_batch_size=4,
            validation_batch_size=3, model_type='tiny_L3', learning_rate=1e-3, checkpoint_interval=1, gpus=1,
            continue_model_dir=md)
    rows = open(os.path.join(md, 'history_csvlog.csv')).read().strip().split('\n')
    assert [r.split(',')[0] for r in rows] == ['epoch', '0', '1', '2']
    assert os.path.exists(os.path.join(md, 'model_checkpoint.03.h5'))


def test_model_state_survives_batch_size_changes(gpu_required):
    """Keras keeps one set of variables whatever batch size is fed.  fit_generator alternates
    train_batch_size and validation_batch_size steps (train.py:408-414), so switching the fed batch size must
    carry the Adam moments / step count and the BatchNorm debias accumulators along (l3_copy_state): a model
    that validates at another batch size between its training steps must end bit-identical to one that never does."""
    mt = 'tiny_L3'
    batches = [o.synthetic_batch(4, seed=80 + k) for k in range(3)]
    vv, va, vl = o.synthetic_batch(3, seed=90)
    ma, _, _ = model.MODELS[mt]()
    mb, _, _ = model.MODELS[mt]()
    for m in (ma, mb):
        m.compile(model.Adam(lr=1e-3), loss='categorical_crossentropy', metrics=['accuracy'])
    mb.set_weights(ma.get_weights())
    for k, (v, a, l) in enumerate(batches):
        la = ma.train_on_batch([v, a], l)
        lb = mb.train_on_batch([v, a], l)
        assert la == lb, k
        assert mb._engine.optimizer_steps() == (k + 1, k + 1)
        before = mb._engine.get_params()
        ev1 = mb.test_on_batch([vv, va], vl)                     # another batch size: the state moves to a second engine
        assert mb._engine.batch == 3 and mb._engine.optimizer_steps() == (k + 1, k + 1)
        after = mb._engine.get_params()
        assert all(np.array_equal(before[n], after[n]) for n in before)
        assert ev1 == mb.test_on_batch([vv, va], vl)
    assert sorted(k[0] for k in mb._engines) == [3, 4] and 3 not in [k[0] for k in ma._engines]
    wa, wb = ma.get_weights(), mb.get_weights()
    assert all(np.array_equal(x, y) for x, y in zip(wa, wb))
    # moving statistics kept their zero-debias history: after 3 updates they are not the last batch's statistics
    mov = [w for (n, _, _), w in zip(ma.param_table(), ma._engine.get_params().values()) if n.endswith('moving_mean')]
    assert all(np.isfinite(x).all() for x in mov)


# ---- size-independent properties at the bench size ---------------------------------------------------------
@pytest.mark.parametrize('dtype,B', [('f32', 64), ('bf16', 128)], ids=['configs2_f32_b64', 'configs4_bf16_b128'])
def test_full_size_properties(gpu_required, dtype, B):
    """BASELINE.json configs[2] (fp32, 64 pairs/GPU) and configs[4] (bf16 operands / fp32 accumulate, 128 pairs/GPU)
    at their real per-GPU size: properties that need no oracle."""
    mt = 'cnn_L3_melspec2'
    eng = _lib.Engine(mt, B, seed=3, dtype=dtype)
    rng = np.random.RandomState(7)
    frm = rng.randint(0, 256, size=(B, 224, 224, 3)).astype(np.uint8)
    pcm = rng.randint(-32768, 32768, size=(B, 1, 48000)).astype(np.int16)
    lab = rng.randint(0, 2, size=(B,))
    labels = np.stack([lab, 1 - lab], 1).astype(np.int32)
    v, a = o.preprocess_video(frm), o.pcm2float(pcm, np.float32)
    # (1) raw (uint8/int16) upload path == float upload path, bit for bit
    eng.upload_batch_raw(frm, pcm, labels)
    eng.step_forward(False)
    l_raw, a_raw, p_raw, z_raw = eng.step_results(True)
    p_f, z_f = eng.forward(v, a, training=False)
    assert np.array_equal(z_raw, z_f)
    # (2) inference mode is per-sample: a batch-4 engine with the same weights gives the same logits
    small = _lib.Engine(mt, 4, seed=3, dtype=dtype)
    small.set_params(eng.get_params())
    p_s, z_s = small.forward(v[8:12], a[8:12], training=False)
    d_small = float(np.abs(z_s - z_f[8:12]).max())
    print('batch-%d vs batch-4 inference logits (%s): max diff %.2e, logit scale %.2f' % (B, dtype, d_small, np.abs(z_f).max()))
    # bf16: a last-bit fp32 difference ahead of an operand rounding can flip it (2^-9 relative on one element)
    assert d_small < (1e-4 if dtype == 'f32' else 5e-3 * max(1.0, float(np.abs(z_f).max())))
    small.close()
    # (3) staged step == monolithic step, and the step is deterministic (bit-identical)
    W0 = eng.get_params()
    eng.upload_batch(v, a, labels.astype(np.float32))
    eng.step_resident(1e-4)
    l1, a1 = eng.step_results()
    W1 = eng.get_params()
    eng.set_params(W0)
    eng.reset_optimizer()
    eng.step_forward(True)
    for b in range(1, eng.bucket_count()):
        eng.step_backward_bucket(b)
    eng.step_update(1e-4, 1.0)
    l2, a2 = eng.step_results()
    W2 = eng.get_params()
    assert l1 == l2 and a1 == a2
    assert all(np.array_equal(W1[k], W2[k]) for k in W1)
    # (4) buckets tile the gradient arena exactly once
    ptr, n = eng.grad_arena()
    rs = [eng.bucket_range(b) for b in range(eng.bucket_count())]
    assert rs[0][0] == 0 and all(rs[i][0] + rs[i][1] == rs[i + 1][0] for i in range(len(rs) - 1)) and rs[-1][0] + rs[-1][1] == n
    assert n >= 9508746
    # (5) repeated steps stay finite (he_normal init saturates the softmax on random inputs, and
    #     keras' probability clipping then zeroes those samples' gradients, so no monotone claim)
    losses = []
    for _ in range(4):
        eng.step_resident(1e-3)
        losses.append(eng.step_results()[0])
    assert np.isfinite(losses).all()
    eng.close()


def test_rccl_world1_trainer_equals_resident_step(gpu_required):
    """torch.distributed 'nccl' (= RCCL) with world size 1: the bucketed all-reduce path must
    alias the engine's gradient arena zero-copy and reproduce the monolithic step."""
    import torch
    import torch.distributed as dist
    from l3embedding_amd.training_utils import DataParallelTrainer
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        mt, B = 'tiny_L3', 4
        v, a, l = o.synthetic_batch(B, seed=71)
        ts = torch.cuda.Stream(device=0)
        assert ts.cuda_stream != 0
        # (rank-local moving statistics: `tr.world = 2` below fakes a second rank that does not exist -- its slot of the gathered
        # BatchNorm statistics would be zeros)
        e1 = _lib.Engine(mt, B, seed=5, stream=ts.cuda_stream, global_batch=B, dp_moving='rank_local')
        e2 = _lib.Engine(mt, B, seed=5)
        e2.set_params(e1.get_params())
        with pytest.raises(ValueError):
            DataParallelTrainer(e1, 0, 2, 0)          # N > 1 without the engine's stream is refused
        tr = DataParallelTrainer(e1, 0, 1, 0, stream=ts)
        assert tr.staged is None and tr.flat.data_ptr() == e1.grad_arena()[0]
        tr.world = 2                      # force the all-reduce code path (sum over one rank)
        e1.upload_batch(v, a, l)
        tr.step(1e-3)
        la, _ = e1.step_results()
        lb, _ = e2.train_step(v, a, l, 1e-3)
        assert la == lb
        Wa, Wb = e1.get_params(), e2.get_params()
        assert all(np.array_equal(Wa[k], Wb[k]) for k in Wa)
        e1.close()
        e2.close()
    finally:
        dist.destroy_process_group()


def test_native_rccl_world1_step_equals_resident_step(gpu_required):
    """RCCL behind the C ABI (l3_comm_unique_id / l3_comm_init / l3_step_dp, SURVEY 8(b),(e)): with one rank the
    per-bucket ncclAllReduce on the communicator's stream is an identity, so the data-parallel step must
    reproduce l3_step_resident bit for bit -- which checks the event ordering between the engine's two
    streams, the communicator's stream and Adam.  Also the small host all-reduce used for logging/barriers."""
    mt, B = 'cnn_L3_melspec2', 2
    v, a, l = o.synthetic_batch(B, seed=71)
    e1 = _lib.Engine(mt, B, seed=5, global_batch=B)
    e2 = _lib.Engine(mt, B, seed=5)
    e2.set_params(e1.get_params())
    with pytest.raises(_lib.L3Error, match='before l3_comm_init'):
        e1.step_dp(1e-3)
    uid = _lib.comm_unique_id()
    assert len(uid) == 128
    e1.comm_init(uid, 1, 0)
    info = e1.comm_info()
    assert info['world'] == 1 and info['rank'] == 0 and 'rccl' in info['library'].lower()
    with pytest.raises(_lib.L3Error, match='already initialised'):
        e1.comm_init(uid, 1, 0)
    e1.upload_batch(v, a, l)
    e2.upload_batch(v, a, l)
    for _ in range(3):
        e1.step_dp(1e-3)
        e2.step_resident(1e-3)
        assert e1.step_results() == e2.step_results()
    # the deferred results summed over the communicator's ranks (one rank: the same values), read after the next step is queued
    with pytest.raises(_lib.L3Error, match='before l3_comm_init'):
        e2.results_enqueue(0, reduce=True)
    seen = []
    for k in range(3):
        e1.step_dp(1e-3)
        e1.results_enqueue(k & 1, reduce=True)
        e2.step_resident(1e-3)
        if k:
            seen.append(e1.results_wait((k - 1) & 1))
        want = e2.step_results()
        if k:
            assert seen[-1] == prev
        prev = want
    assert e1.results_wait(2 & 1) == prev
    Wa, Wb = e1.get_params(), e2.get_params()
    assert all(np.array_equal(Wa[k], Wb[k]) for k in Wa)
    assert e1.comm_allreduce([3.5, -1.0, 2.0 ** 40], 'sum') == [3.5, -1.0, 2.0 ** 40]
    assert e1.comm_allreduce([3.5, -1.0], 'max') == [3.5, -1.0]
    e1.comm_destroy()
    e1.comm_destroy()                      # idempotent
    e1.close()
    e2.close()


def _run_fake_dp(mt, B, steps, world, fault=None):
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(here, 'fake_rccl', 'libfake_rccl.so')
    assert os.path.exists(lib), 'tests/fake_rccl/libfake_rccl.so missing: __graft_entry__.build() makes it'
    env = dict(os.environ, L3_RCCL_LIB=lib, L3_DEBUG_KNOBS='1')
    env.pop('L3_DP_FAULT', None)
    if fault:
        env['L3_DP_FAULT'] = str(fault)
    if fault == 1:
        env['FAKE_RCCL_DELAY_US'] = '0'       # the early collective must really run early, not be delayed past the backward
    r = subprocess.run([sys.executable, os.path.join(here, 'dp_fake_worker.py'), mt, str(B), str(steps), str(world)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = r.stdout.decode(errors='replace')
    assert r.returncode == 0, text[-3000:]
    return json.loads([ln for ln in text.splitlines() if ln.startswith('RESULT ')][-1][7:])


@pytest.mark.gpu
def test_dp_event_ordering_with_a_fake_collective(gpu_required):
    """The data-parallel ORDERING (training_utils.py:141-170 semantics: every replica's gradient is in the sum before
    the optimizer reads it), tested on one GPU.  At world 1 a real all-reduce is an identity and would hide a
    collective launched before its bucket's backward, or an Adam launched before the last collective.  Here
    libl3hip binds a double of librccl (tests/fake_rccl) whose all-reduce is `x *= world` behind a 300-us spinning
    kernel on the communicator stream; the engine runs with global batch = world * B, so its loss gradients carry
    1/(world*B) and the "reduced" gradients must equal those of the plain step at global batch B bit for bit (powers
    of two) -- for every tensor, after three steps, weights included.
    Negative controls (debug-gated fault injection in l3_step_dp): reducing each bucket before its backward, and
    Adam without the wait on the communicator stream, must both be SEEN by the same comparison."""
    mt, B, steps, world = 'cnn_L3_melspec2', 2, 3, 2
    ok = _run_fake_dp(mt, B, steps, world)
    # gradients of the last step and every weight after three steps, bit for bit
    assert ok['param_mismatch'] == [] and ok['grad_mismatch'] == [] and ok['n_tensors'] > 100 and ok['grad_nonzero'] > 50, ok
    assert ok['collectives'] >= steps * ok['buckets'] and ok['buckets'] == 9
    assert ok['host_sum'] == [3.0, -4.0] and ok['host_max'] == [1.5, -2.0]        # the double's sum is world * x
    early = _run_fake_dp(mt, B, steps, world, fault=1)
    noadamwait = _run_fake_dp(mt, B, 1, world, fault=2)
    print('fault 1 (every bucket reduced before its backward): %d of %d tensors differ; fault 2 (Adam without the wait on the '
          'communicator stream): %d differ' % (len(early['param_mismatch']), early['n_tensors'], len(noadamwait['param_mismatch'])))
    assert len(early['param_mismatch']) > 50 and len(early['grad_mismatch']) > 50, early      # backward overwrote the "reduced"
                                                                                              # buckets: half-size gradients
    assert len(noadamwait['param_mismatch']) > 0, noadamwait   # Adam read buckets still on the wire


@pytest.mark.gpu
@pytest.mark.parametrize('peer,what', [('4:0', 'bucket order'), ('6:1', 'dp_moving'), ('1:99', 'number of gradient buckets'), (None, None)])
def test_comm_init_refuses_ranks_that_were_configured_differently(gpu_required, peer, what):
    """Every rank must issue the same sequence of collectives (training_utils.py:141-170 is ONE graph in the reference; here it is N
    processes): l3_comm_init fixes the bucket order and has the ranks compare it, the bucket count, dp_moving and model / precision with
    one MAX all-reduce of (x, -x).  The collective double plays a peer that reports another value in one slot
    (FAKE_RCCL_MAX_PEER): initialisation fails with L3_ECOMM and a message naming what differs, and leaves no communicator behind;
    without such a peer it succeeds.  (Slots: 0-3 = order, buckets, dp_moving, model / precision; 4-7 = their negatives.)"""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, L3_RCCL_LIB=os.path.join(here, 'fake_rccl', 'libfake_rccl.so'), L3_DEBUG_KNOBS='1', FAKE_RCCL_DELAY_US='0')
    env.pop('FAKE_RCCL_MAX_PEER', None)
    if peer is not None:
        env['FAKE_RCCL_MAX_PEER'] = peer
    r = subprocess.run([sys.executable, os.path.join(here, 'dp_fake_worker.py'), 'disagree'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=300)
    text = r.stdout.decode(errors='replace')
    assert r.returncode == 0, text[-2000:]
    res = json.loads([ln for ln in text.splitlines() if ln.startswith('RESULT ')][-1][7:])
    if peer is None:
        assert res['error'] is None and res['no_comm_left'] is False      # a communicator exists: the step runs
    else:
        assert res['error'] is not None and 'ranks disagree on' in res['error'] and what in res['error'], res
        assert res['no_comm_left'] is True


@pytest.mark.gpu
@pytest.mark.parametrize('zero_debias', [1, 0], ids=['zero_debias', 'plain_ema'])
def test_dp_moving_statistics_take_one_update_per_replica(gpu_required, zero_debias):
    """multi_gpu_model calls the template model once per replica (training_utils.py:141-157), so every BatchNormalization
    (vision_model.py:124-187, audio_model.py:370-433) updates its shared moving mean / variance `gpus` times per step.  The engine
    (l3_config.dp_moving = L3_DP_MOVING_REPLICAS, the default) gathers every rank's batch statistics and applies them in replica
    order.  Here: tiny_L3, "world 2" through the collective double (both replicas hold this rank's shard), three steps on three
    different batches, against the oracle's virtual-rank step (oracle.dp_train_step on the shard repeated twice); dp_moving =
    rank_local against one update per step.  Learning rate 0: the statistics then depend on the data alone -- with a live
    optimizer the convolution biases in front of a BatchNorm (whose true gradient is zero) take +-lr steps on rounding noise
    and move the batch MEANS by as much, which says nothing about the moving-average rule under test."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(here, 'fake_rccl', 'libfake_rccl.so')
    mt, B, steps, world = 'tiny_L3', 3, 3, 2
    env = dict(os.environ, L3_RCCL_LIB=lib, L3_DEBUG_KNOBS='1', FAKE_RCCL_DELAY_US='50')
    env.pop('L3_DP_FAULT', None)
    r = subprocess.run([sys.executable, os.path.join(here, 'dp_fake_worker.py'), 'moving', mt, str(B), str(steps), str(world), str(zero_debias)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = r.stdout.decode(errors='replace')
    assert r.returncode == 0, text[-3000:]
    res = json.loads([ln for ln in text.splitlines() if ln.startswith('RESULT ')][-1][7:])
    worst = {}
    for mode in ('replicas', 'rank_local'):
        P = o.init_params(mt, seed=5)
        eng = _lib.Engine(mt, B, seed=5)
        P.update(eng.get_params())                    # the engine's own initial weights
        eng.close()
        adam, bn = o.AdamState(), o.BNMovingState(zero_debias=bool(zero_debias))
        for k in range(steps):
            v, a, l = o.synthetic_batch(B, seed=71 + k)
            o.dp_train_step(mt, P, adam, bn, np.concatenate([v, v]), np.concatenate([a, a]), np.concatenate([l, l]), 0.0, world, moving=mode)
        assert res[mode + '_steps'] == [steps, steps * (world if mode == 'replicas' else 1)]
        got = res[mode]
        assert len(got) >= 12
        worst[mode] = max(relerr(np.asarray(got[k], np.float32), P[k]) for k in got)
        assert worst[mode] < 2e-5, (mode, worst)
    # and the two modes really differ (else the test could not tell them apart): the statistics move across the steps
    d = max(relerr(np.asarray(res['replicas'][k]), np.asarray(res['rank_local'][k])) for k in res['replicas'])
    print('moving statistics after %d data-parallel steps, %s: replicas within %.1e of the oracle, rank_local within %.1e; the modes differ by %.1e'
          % (steps, 'zero-debias' if zero_debias else 'plain EMA', worst['replicas'], worst['rank_local'], d))
    assert d > 1e-3          # measured 4.1e-3 (zero-debias: both modes are unbiased averages, they differ in the weights) / 0.97 (plain EMA)


@pytest.mark.gpu
@pytest.mark.parametrize('footprint', ['thin', 'ring'])
def test_slow_collectives_hide_behind_backward(gpu_required, footprint):
    """OVERLAP, not just order: with the double of librccl taking 300 us per bucket on the communicator stream
    (9 buckets = 2.7 ms of "wire" per step), the optimizer of a data-parallel step of the full model at 64 pairs may wait at most
    0.75 ms for the wire once backward is done (measured 0.45-0.64) (l3_comm_timing: the last bucket's own 0.3 ms cannot hide -- it becomes ready when
    backward ends), and the step may cost at most 1.0 ms more than the plain step on the same engine.  It fails if the collectives
    serialise behind backward or cannot start beside the persistent Winograd grids -- and it did: with the buckets enqueued in
    arena order (vision 4..1, then audio 4..1) every audio bucket waited on the communicator stream behind the LAST vision bucket:
    1.4-1.5 ms exposed, +2.0 ms per step (L3_DP_ARENA_ORDER=1 restores that order; profiles/r05_dp_overlap.txt).  The ~0.35 ms the
    step grows by beyond the exposed wait are the same with and without persistent convolution grids (L3_WINO_PERSIST=0: +0.84
    against +0.89): nine spinning kernels and eighteen more launches beside a chip that is never idle.
    footprint 'ring' (round 6): the collective's stand-in is 32 workgroups x 512 threads x 64 KiB of LDS -- the CU
    footprint of a real RCCL ring kernel, which cannot share a CU with a 150-KiB convolution workgroup -- each holding its CU for the
    300 us.  The convolutions of a data-parallel step then take their tile blocks from work counters (ConvGeom::dynamic): a workgroup
    the collective keeps from starting costs its share, not a second round.  Same bounds (profiles/r06_dp_footprint.txt)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(here, 'fake_rccl', 'libfake_rccl.so')
    assert os.path.exists(lib)
    env = dict(os.environ, L3_RCCL_LIB=lib, L3_DEBUG_KNOBS='1', FAKE_RCCL_DELAY_US='300', GPU_MAX_HW_QUEUES='8')
    env.pop('L3_DP_FAULT', None)
    env.pop('L3_W4_DYNAMIC', None)
    if footprint == 'ring':
        env.update(FAKE_RCCL_BLOCKS='32', FAKE_RCCL_THREADS='512', FAKE_RCCL_LDS_KB='64')
    r = subprocess.run([sys.executable, os.path.join(here, 'dp_fake_worker.py'), 'overlap', 'cnn_L3_melspec2', '64', '20', '2'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = r.stdout.decode(errors='replace')
    assert r.returncode == 0, text[-3000:]
    res = json.loads([ln for ln in text.splitlines() if ln.startswith('RESULT ')][-1][7:])
    ct = res['comm_timing']
    print('[%s] plain step %.2f ms, data-parallel step with 9 x 300 us collectives %.2f ms (+%.2f); in-library timing: exposed %.3f ms, '
          'span %.2f ms, buckets %s' % (footprint, res['plain_ms'], res['dp_ms'], res['dp_ms'] - res['plain_ms'], ct['exposed_ms'], ct['span_ms'],
                                        ['%.2f' % x for x in ct['bucket_ms']]))
    assert sum(ct['bucket_ms']) >= 9 * 0.28          # the wire really was slow
    # Bounds relative to the collective's own duration (absolute milliseconds flake across boxes): what backward cannot
    # hide is the LAST bucket's collective plus the BatchNorm-statistics gather queued behind bucket 0 -- at most ~2 collectives'
    # worth once everything else overlaps (measured 0.45-0.64 ms at 0.31-0.34 ms per collective); serialised buckets would expose
    # 9 of them.  The step may grow by the exposed wait plus the launches of twenty more kernels beside a chip that is never idle:
    # measured +0.82 ... +1.23 ms over eight boxes at 0.57-0.64 ms exposed; collectives that serialise behind backward cost 9 x per =
    # +2.8 ms and collectives parked behind the last vision bucket +2.0 ms (round 5) -- both stay outside the bound.
    per = sorted(ct['bucket_ms'])[len(ct['bucket_ms']) // 2]       # (median: a collective whose workgroups wait for CUs stretches)
    assert ct['exposed_ms'] <= 2.5 * per, res
    assert res['dp_ms'] - res['plain_ms'] <= ct['exposed_ms'] + 3.0 * per, res


@pytest.mark.gpu
@pytest.mark.parametrize('mt,B', [('tiny_L3', 5), ('cnn_L3_melspec2', 2)])
def test_tower_overlap_is_bit_identical(gpu_required, mt, B):
    """The audio tower on the engine's side stream (default) and both towers serialised on one
    stream are the same arithmetic: three training steps must agree bit for bit."""
    v, a, l = o.synthetic_batch(B, seed=123)
    e1 = _lib.Engine(mt, B, seed=9)
    e2 = _lib.Engine(mt, B, seed=9)
    e2.set_params(e1.get_params())
    e2.set_tower_overlap(False)
    for _ in range(3):
        la, _ = e1.train_step(v, a, l, 1e-3)
        lb, _ = e2.train_step(v, a, l, 1e-3)
        assert la == lb
    pa, lga = e1.forward(v, a, training=False)
    pb, lgb = e2.forward(v, a, training=False)
    assert np.array_equal(lga, lgb) and np.array_equal(pa, pb)
    Wa, Wb = e1.get_params(), e2.get_params()
    assert all(np.array_equal(Wa[k], Wb[k]) for k in Wa)
    e1.close()
    e2.close()


# ---- mixed precision (BASELINE.json configs[4]) -------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(1, 17, 13, 64, 128), (2, 5, 5, 64, 64), (1, 6, 6, 256, 512), (3, 32, 24, 128, 192),
                                   (1, 33, 31, 64, 64), (2, 9, 11, 16, 64)])
def test_conv2d_bf16_operands(gpu_required, shape):
    """L3_DTYPE_BF16 convolution = conv(bf16(x), bf16(w)) with fp32 accumulation: against the oracle's
    emulation the only difference is summation order (bf16 products are exact in fp32).  A geometry the
    mixed-precision rule does not cover (last case: Cin = 16) must stay fp32-exact."""
    n, h, w, ci, co = shape
    rng = np.random.RandomState(sum(shape))
    x = rng.randn(n, h, w, ci).astype(np.float32)
    wt = (rng.randn(3, 3, ci, co) / np.sqrt(9 * ci)).astype(np.float32)
    b = rng.randn(co).astype(np.float32)
    dy = rng.randn(n, h, w, co).astype(np.float32)
    x64, w64, b64, dy64 = (t.astype(np.float64) for t in (x, wt, b, dy))
    with o.mixed_precision('bf16'):
        y_ref = o.conv2d_fwd(x64, w64, b64, 'same')
        dx_ref, dw_ref, db_ref = o.conv2d_bwd(x64, w64, dy64, 'same')
    y = _lib.op_conv2d_fwd(x, wt, b, True, dtype='bf16')
    dx, dw, db = _lib.op_conv2d_bwd(x, wt, dy, True, dtype='bf16')
    assert relerr(y, y_ref) < 5e-6 and relerr(dx, dx_ref) < 5e-6 and relerr(dw, dw_ref) < 5e-6 and relerr(db, db_ref) < 5e-6
    if ci % 64 == 0:      # and it really is the rounded computation, not the fp32 one
        assert relerr(y, o.conv2d_fwd(x64, w64, b64, 'same')) > 1e-4


@pytest.mark.gpu
def test_bf16_training_step_matches_mixed_precision_oracle(gpu_required):
    """Full cnn_L3_melspec2 step in L3_DTYPE_BF16 mode against the oracle run with the same operand
    rounding (oracle.mixed_precision).

    Tolerances: rounding to bf16 is discontinuous, so two correct implementations that differ in the
    last fp32 bit *before* a rounding step drift apart layer by layer (err -> sqrt(err / 128), fixed
    point ~ 2^-7) until they are about as far from each other as bf16 is from fp32.  So: (1) the first
    mixed-precision conv of each tower, whose inputs are still fp32-identical, must match the bf16
    oracle tightly and must NOT match the fp32 oracle; (2) logits / loss must be within the
    bf16-vs-fp32 distance of the bf16 oracle; (3) the update is sane."""
    mt, B = 'cnn_L3_melspec2', 2
    P = o.init_params(mt, seed=77)
    v, a, l = o.synthetic_batch(B, seed=78)
    eng = _lib.Engine(mt, B, dtype='bf16')
    eng.set_params(P)
    with o.mixed_precision('bf16'):
        ref = o.forward(mt, P, v, a, True, np.float64, want_taps=True)
        Pn = {k: np.array(x, copy=True) for k, x in P.items()}
        adam, bn = o.AdamState(), o.BNMovingState()
        r = o.train_step(mt, Pn, adam, bn, v, a, l, 1e-4, np.float64)
    ref32 = o.forward(mt, P, v, a, True, np.float64, want_taps=True)
    probs, logits = eng.forward(v, a, training=True)
    for name in ('vision_model/conv2d_2', 'audio_model/conv2d_9'):
        tap = name.split('/')[1]
        act = eng.activation(name).reshape(ref['taps'][tap].shape).astype(np.float64)
        # rare rounding flips of single inputs show in the max error; the MEAN error separates cleanly
        mean_err = lambda want: float(np.abs(act - want).mean() / np.abs(want).mean())
        e16, e32 = mean_err(ref['taps'][tap]), mean_err(ref32['taps'][tap])
        # the tap is stored as bfloat16 (oracle rule (2)): a near-tie may land on the neighbouring bfloat16, i.e.
        # up to one spacing (2^-7 of the element) away; everything else is exact
        assert relerr(act, ref['taps'][tap]) < 2.0 ** -7 and e16 < 1e-4, (name, e16)
        assert np.array_equal(o.bf16_round(act), act), name
        assert e32 > 4e-4 and e32 > 5 * e16, (name, e16, e32)           # it is the rounded computation
    d_mp = np.abs(ref['logits'] - ref32['logits']).max()
    scale = np.abs(ref['logits']).max()
    assert d_mp > 1e-3                                                   # the mode is measurably not fp32
    d_gpu = float(np.abs(logits - ref['logits']).max())
    loss, acc = eng.train_step(v, a, l, 1e-4)
    print('bf16 step: |logits - bf16 oracle| %.3e (bf16 oracle vs fp32 oracle %.3e, logit scale %.2f); loss %.6f vs %.6f'
          % (d_gpu, d_mp, scale, loss, r['loss']))
    assert d_gpu < max(2.0 * d_mp, 0.02 * scale)
    assert abs(loss - r['loss']) < 0.05 * max(1.0, abs(r['loss']))
    W = eng.get_params()
    for name, _, trainable, _ in o.param_table(mt):
        if trainable:   # one Adam step moves every weight by ~lr in either implementation
            assert np.abs(W[name] - P[name]).max() < 1.05e-4 + 1e-6 * np.abs(P[name]).max(), name
    eng.close()


# measured on MI355X (profiles/r05_mixed_golden_distances.txt): the bf16 engine against the MIXED-PRECISION oracle at batch 8.
# logits 3.3e-2 (training) / 4.9e-2 (inference) where the mixed oracle is 9.9e-2 / 1.45e-1 from float64; loss 3.2e-4; first
# mixed-precision activations 1.8e-6 / 4.2e-5 of their mean size, the embedding layers 9.6e-3 / 8.6e-3.  GRADIENTS do not get
# closer than the yardstick: bfloat16 stores are discontinuous, so a last-bit difference in an fp32 sum flips a stored element by
# 2^-8 and, downstream, ReLU masks and pool winners -- two correct implementations decorrelate element by element (sampled L2
# 0.12-0.34 per tensor, worst sampled element 2.0 RMS) exactly as the mixed oracle does from float64 (worst 2.9 RMS); the bound
# is that yardstick.
# Round 6: the LOSS bound was 1e-3 on a measured 3.2e-4 -- luck, not a property: the cross-entropy moves by up to 2 max|dlogits| and the
# logits are 3e-2 from the mixed oracle's, so which way eight samples' errors add up is decided by the last bits of the INPUT.  The three
# front-end forms of round 6 (full DFT GEMM, two-GEMM factorisation, one-kernel factorisation: 4.7e-5 / 3.6e-5 / 3.4e-5 dB from the
# float64 front-end, i.e. each at least as close as the one before) give 3.2e-4 / 3.5e-4 / 6.3e-3 here.  The bound is now tied to what it
# follows from: a third of the logits bar.
MIXED_B8 = dict(logits_train=6e-2, logits_eval=9e-2, loss=2e-2, tap_first_mean=1e-4, tap_last_mean=3e-2, grad_l2=0.6)


@pytest.mark.gpu
def test_bf16_training_step_matches_the_mixed_precision_golden_at_batch_8(gpu_required):
    """BASELINE.json configs[4] against a committed vector: tests/golden/cnn_L3_melspec2_b8_bf16.npz is one training step of the
    oracle in mixed-precision mode (bfloat16 operands / fp32 accumulate on the 14 wide 3x3 layers, bfloat16-stored activations and
    data gradients; tests/golden/make_mixed_golden.py) at batch 8, where the BatchNorm statistics are well conditioned and two
    correct implementations stay close -- unlike batch 2 (test_bf16_training_step_matches_mixed_precision_oracle), whose bound
    is the bf16-vs-fp32 distance itself.  Here the engine must be CLOSER to the mixed-precision oracle than that oracle is to the
    float64 one -- by 3x for the logits, by orders of magnitude for the first mixed-precision activation of each tower -- and the
    sampled gradients no further (see MIXED_B8)."""
    z = np.load(os.path.join(GOLDEN, 'cnn_L3_melspec2_b8_bf16.npz'))
    mod = _mod()
    mt, B = str(z['model_type']), int(z['batch'])
    P = mod.perturbed_params(mt, int(z['param_seed']))
    v, a, l = o.synthetic_batch(B, seed=int(z['data_seed']))
    eng = _lib.Engine(mt, B, dtype='bf16')
    eng.set_params(P)
    _, logits_e = eng.forward(v, a, training=False)
    probs, logits = eng.forward(v, a, training=True)
    d_eval = float(np.abs(logits_e - z['eval_logits']).max())
    d_train = float(np.abs(logits - z['train_logits']).max())
    yard = float(z['logits_vs_float64'])
    taps = {}
    for name, tap in (('vision_model/conv2d_2', 'conv2d_2'), ('audio_model/conv2d_9', 'conv2d_9'),
                      ('vision_model/vision_embedding_layer', 'vision_embedding_layer'), ('audio_model/audio_embedding_layer', 'audio_embedding_layer')):
        act = eng.activation(name).astype(np.float64).ravel()
        idx = mod.sample_idx(tap, act.size, 4096)
        taps[tap] = float(np.abs(act[idx] - z['tap:' + tap]).mean() / float(z['tapabs:' + tap]))
    loss, acc = eng.train_step(v, a, l, float(z['lr']))
    d_loss = abs(loss - float(z['loss']))
    G = eng.get_grads()
    worst = dict(err=0.0, l2=0.0, yard_err=0.0)
    rows = []
    for n, _, tr in eng.param_table():
        if not tr or float(z['gnorm:' + n]) < 1e-7:
            continue
        got = G[n].astype(np.float64)
        if n.endswith('/kernel'):
            got = got - 2 * o.L2_WEIGHT * P[n].astype(np.float64)
        idx = mod.sample_idx(n, got.size)
        ref, gnorm = z['gsamp:' + n], float(z['gnorm:' + n])
        err, nerr = mod.grad_metrics(got, ref, gnorm, idx)
        l2 = float(np.sqrt(((got.ravel()[idx] - ref) ** 2).sum() / ((ref ** 2).sum() + 1e-300)))
        y64 = float(z['gd64:' + n][0]) if 'gd64:' + n in z.files else float('nan')
        rows.append((n, err, l2, y64))
        if got.size > 1:                 # (single-element tensors: one cancelling sum over every pixel, see the fp32 golden test)
            worst['err'], worst['l2'] = max(worst['err'], err), max(worst['l2'], l2)
            worst['yard_err'] = max(worst['yard_err'], y64)
    print('mixed-precision golden, batch 8: |logits - mixed oracle| training %.3e inference %.3e (mixed oracle vs float64: %.3e); loss %.6f vs %.6f'
          % (d_train, d_eval, yard, loss, float(z['loss'])))
    print('   activations, mean |difference| / mean |value|: ' + ', '.join('%s %.2e' % kv for kv in taps.items()))
    print('   gradients: worst sampled err/rms %.3e, sampled L2 %.3e   (mixed oracle vs float64: err/rms %.3e)' % (worst['err'], worst['l2'], worst['yard_err']))
    for r in sorted(rows, key=lambda r: -r[1])[:6]:
        print('      %-50s err/rms %.3e  L2 %.3e  (yardstick %.3e)' % r)
    B8 = MIXED_B8
    assert d_train < B8['logits_train'] and d_train < 0.7 * yard, (d_train, yard)
    assert d_eval < B8['logits_eval'] and d_eval < 0.7 * float(z['eval_logits_vs_float64']), (d_eval, float(z['eval_logits_vs_float64']))
    assert d_loss < B8['loss'], (loss, float(z['loss']))
    assert acc == pytest.approx(float(z['acc']))
    assert taps['conv2d_2'] < B8['tap_first_mean'] and taps['conv2d_9'] < B8['tap_first_mean'], taps
    assert max(taps['vision_embedding_layer'], taps['audio_embedding_layer']) < B8['tap_last_mean'], taps
    assert worst['err'] < worst['yard_err'] and worst['l2'] < B8['grad_l2'], rows
    eng.close()


# Training-level acceptance of BASELINE configs[4] (VERDICT r05 #3a): the reference is fp32 throughout (audio_model.py:363,
# vision_model.py:123); the mixed-precision rules are the build's own, so what they must be held to is the build's own fp32
# engine on the same weights and data, at the shard size the configuration runs (128 pairs per GPU).
# Measured (profiles/r06_bf16_vs_f32_gradients.txt): per-tensor cosine 0.84-0.95, whole gradient 0.955 -- the 0.95 per tensor VERDICT
# r05 proposed is NOT reached, and that is a finding about rules (2)/(3), recorded there (bfloat16-stored activations / data gradients
# toggle ReLU-mask and arg-max paths; the noise grows down the backward pass), not a bug: two fp32 roundings of the same batch agree
# to 1.0000, two fp32 MINIBATCHES are nearly orthogonal per tensor (-0.36...+0.30; whole gradient 0.39), and thirty training steps
# stay within 1.7 % in the loss.  The bars are those measurements with margin, each tied to a yardstick taken in the same test.
BF16_GRAD_COSINE_MIN = 0.80          # per tensor with more than three elements (measured min 0.837)
BF16_GRAD_COSINE_WHOLE = 0.93        # the whole gradient (measured 0.955)
BF16_OVER_MINIBATCH = 0.40           # ... and at least this far above the cosine of two fp32 minibatches on that tensor (measured >= 0.55)
F32_FLOOR = 0.9999                   # fp32 F(4x4,3x3) against fp32 F(2x2,3x3) on the same batch: what fp32 leaves undetermined
BF16_TRAJ_REL = 0.05                 # loss of every one of 30 steps, relative to the fp32 engine's (measured <= 1.7e-2)


@pytest.mark.gpu
def test_bf16_engine_trains_like_the_fp32_engine_at_the_shard_size(gpu_required):
    """(a) one step at batch 128 on seeded, perturbed weights (tests/golden/make_golden.py perturbed_params) with a live head
    (dense_2/kernel / 64 as in bench.py: every sample has a loss gradient): per-tensor COSINE between the bf16 engine's gradient and
    the fp32 engine's -- the whole table is printed -- beside two yardsticks on the same weights: the fp32 engine under its other
    convolution algorithm (rounding only) and the fp32 engine on ANOTHER batch (minibatch sampling noise).  (b) thirty training
    steps over a fixed cycle of four batches, both engines from the same weights: the bf16 engine's loss within 5 % of the fp32
    engine's at EVERY step."""
    mt, B = 'cnn_L3_melspec2', 128
    mod = _mod()
    P = mod.perturbed_params(mt, 101)
    P['dense_2/kernel'] = (P['dense_2/kernel'] / np.float32(64)).astype(np.float32)
    batches = [o.synthetic_batch(B, seed=500 + k) for k in range(4)]

    def grads(dtype, conv, batch):
        e = _lib.Engine(mt, B, seed=0, dtype=dtype, fp32_conv=conv)
        e.set_params(P)
        v, a, l = batch
        probs, _ = e.forward(v, a, training=True)
        pt = (probs.astype(np.float64) * l).sum(axis=1)                 # outside [1e-7, 1 - 1e-7] keras' clip zeroes the gradient (train.py:282-284)
        assert float(np.mean((pt > 1e-7) & (pt < 1 - 1e-7))) > 0.9
        e.upload_batch(v, a, l)
        e.step_forward(True)
        for b in range(1, e.bucket_count()):
            e.step_backward_bucket(b)
        e.sync()
        g = e.get_grads()
        e.step_update(0.0, 1.0)                        # close the step without moving anything
        e.close()
        out = OrderedDict()
        for n, x in g.items():
            x = x.astype(np.float64).ravel()
            if n.endswith('/kernel'):                  # the L2 term is identical in every engine: compare the data term
                x = x - 2 * o.L2_WEIGHT * P[n].astype(np.float64).ravel()
            out[n] = x
        return out

    def cos(x, y):
        nx, ny = np.linalg.norm(x), np.linalg.norm(y)
        return float(x @ y / (nx * ny)) if nx > 0 and ny > 0 else float('nan')

    g32, g16 = grads('f32', 'f4x4', batches[0]), grads('bf16', 'f4x4', batches[0])
    g22, g32b = grads('f32', 'f2x2', batches[0]), grads('f32', 'f4x4', batches[1])
    rows = [(n, g32[n].size, cos(g32[n], g16[n]), cos(g32[n], g22[n]), cos(g32[n], g32b[n])) for n in g32]
    print('gradient of batch A, fp32 engine, cosine per tensor with: the bf16 engine | the fp32 F(2x2,3x3) engine | the fp32 engine on batch B')
    for n, size, c16, c22, cb in sorted(rows, key=lambda r: r[2] if r[2] == r[2] else 9):
        print('   %-54s %9d  %8.4f %8.4f %8.4f' % (n, size, c16, c22, cb))
    whole = lambda g: np.concatenate([g[n] for n in g32])
    w16, w22, wb = cos(whole(g32), whole(g16)), cos(whole(g32), whole(g22)), cos(whole(g32), whole(g32b))
    print('   whole gradient: bf16 %.4f, fp32 F(2x2,3x3) %.4f, another minibatch %.4f' % (w16, w22, wb))
    # judged: tensors with a real gradient -- not the convolution biases in front of a BatchNorm (zero true gradient: rounding noise in
    # both engines) and not the 1- / 3-element input-BatchNorm tensors (one cancelling sum over every pixel, see the fp32 golden test)
    judged = [r for r in rows if r[1] > 3 and not (r[0].endswith('/bias') and not r[0].startswith('dense'))]
    assert len(judged) >= 50
    assert [r for r in judged if not r[3] >= F32_FLOOR] == []                  # the yardstick is a yardstick
    assert [r for r in judged if not r[2] >= BF16_GRAD_COSINE_MIN] == []
    assert [r for r in judged if not r[2] - r[4] >= BF16_OVER_MINIBATCH] == []
    assert w16 >= BF16_GRAD_COSINE_WHOLE and w22 >= F32_FLOOR and w16 - wb >= BF16_OVER_MINIBATCH
    # ---- (b) loss trajectory ----
    losses = {}
    for dt in ('f32', 'bf16'):
        e = _lib.Engine(mt, B, seed=0, dtype=dt)
        e.set_params(P)
        losses[dt] = [e.train_step(*batches[k % 4], 1e-4)[0] for k in range(30)]
        e.close()
    rel = [abs(x - y) / abs(y) for x, y in zip(losses['bf16'], losses['f32'])]
    print('30 steps on a 4-batch cycle: fp32 loss %.4f -> %.4f, bf16 %.4f -> %.4f; worst relative distance %.3e (step %d)'
          % (losses['f32'][0], losses['f32'][-1], losses['bf16'][0], losses['bf16'][-1], max(rel), int(np.argmax(rel))))
    assert losses['f32'][-1] < 0.8 * losses['f32'][0]           # it trains
    assert max(rel) <= BF16_TRAJ_REL, rel


@pytest.mark.gpu
def test_batch_beyond_2gib_tensors_matches_replicated_small_batch(gpu_required):
    """At 192 pairs/GPU the block-1 activations exceed 2 GiB, the reach of the 32-bit buffer offsets the
    fast kernels use, so those layers run as several launches over sample ranges (Winograd, wgrad with
    the ranges as extra split-K partials, bf16).  Property: a batch made of three copies of a 64-pair
    batch has the same BatchNorm statistics, logits and (mean-loss) gradients as the 64-pair batch."""
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 120e9:
        pytest.skip('needs ~90 GB of HBM')
    mt, B, R = 'cnn_L3_melspec2', 64, 3
    v, a, l = o.synthetic_batch(B, seed=4)
    e1 = _lib.Engine(mt, B, seed=3)
    P = e1.get_params()
    _, lg1 = e1.forward(v, a, training=True)
    loss1, _ = e1.train_step(v, a, l, 1e-4)
    G1 = e1.get_grads()
    e1.close()
    e3 = _lib.Engine(mt, B * R, seed=3)
    e3.set_params(P)
    v3, a3, l3 = (np.concatenate([t] * R, axis=0) for t in (v, a, l))
    _, lg3 = e3.forward(v3, a3, training=True)
    assert np.abs(lg3[:B] - lg1).max() < 2e-4 and np.abs(lg3[2 * B:] - lg1).max() < 2e-4
    loss3, _ = e3.train_step(v3, a3, l3, 1e-4)
    assert abs(loss3 - loss1) < 1e-4 * max(1.0, abs(loss1))
    G3 = e3.get_grads()
    e3.close()
    for name in (n for n in G1 if not n.endswith('/bias') or n.startswith('dense')):   # (see below for the biases)
        # Loose on purpose: this network's backward pass amplifies fp32 round-off (16 batch-norm backward
        # stages; the same comparison at batch 8 vs 24 differs by up to 5 % while the float64 oracle
        # satisfies the property to 1e-13), and the forward half above is the tight check.  A broken
        # fallback kernel shows up as an O(1) error.  Conv biases in front of a BatchNorm have a
        # mathematically zero gradient (pure round-off) and are skipped.
        tol = 0.15 * np.abs(G1[name]).max() + 1e-5
        if G1[name].size == 1:
            # gamma / beta of the audio tower's single-channel input BatchNorm: scaling or shifting the one input channel
            # scales / shifts every output of the first convolution, which the BatchNorm behind it removes again -- the
            # gradient is zero up to epsilon and border effects (6e-5 here, sums of terms ~1e-2), i.e. mostly round-off
            # like the biases; what is checked is that it stays at that level
            tol = 1.0 * np.abs(G1[name]).max() + 1e-5
        assert np.abs(G3[name] - G1[name]).max() < tol, name


@pytest.mark.gpu
def test_staged_raw_batches_equal_synchronous_uploads(gpu_required):
    """l3_stage_batch_raw (next batch over the copy stream while a step runs, adopted by the following
    step) must give the same trajectory as synchronous uploads of the same batches -- and
    fit_generator's pipelined loop the same losses as train_on_batch in a plain loop."""
    from l3embedding_amd import model as lm
    mt, B = 'tiny_L3', 6
    rng = np.random.RandomState(3)
    batches = []
    for _ in range(5):
        vid = rng.randint(0, 256, size=(B, 224, 224, 3)).astype(np.uint8)
        aud = rng.randint(-32768, 32768, size=(B, 1, 48000)).astype(np.int16)
        lab0 = rng.randint(0, 2, B)
        batches.append((vid, aud, np.stack([lab0, 1 - lab0], 1).astype(np.int32)))
    e1, e2 = _lib.Engine(mt, B, seed=2), _lib.Engine(mt, B, seed=2)
    e2.set_params(e1.get_params())
    e2.upload_batch_raw(*batches[0])
    for k, b in enumerate(batches):
        e1.upload_batch_raw(*b)
        e1.step_resident(1e-3)
        la = e1.step_results()
        e2.step_resident(1e-3)                       # adopts what was staged (or the first explicit upload)
        if k + 1 < len(batches):
            e2.stage_batch_raw(*batches[k + 1])      # while step k is still running
        lb = e2.step_results()
        assert la == lb, k
    Wa, Wb = e1.get_params(), e2.get_params()
    assert all(np.array_equal(Wa[n], Wb[n]) for n in Wa)
    e1.close()
    e2.close()
    # the Keras-protocol loop
    def gen():
        for v, a, l in batches:
            yield [v, a], l
    losses = []
    class Rec(object):
        batch_hooks_are_passive = True          # they only record `logs`: the pipelined order is allowed (model._batch_hooks_passive)

        def on_train_begin(self, logs): pass
        def on_train_end(self, logs): pass
        def on_epoch_begin(self, e, logs): pass
        def on_epoch_end(self, e, logs): pass
        def on_batch_begin(self, b, logs): pass
        def on_batch_end(self, b, logs): losses.append(logs['loss'])
    m1, _, _ = lm.MODELS[mt]()
    m1.compile(lm.Adam(lr=1e-3), loss='categorical_crossentropy', metrics=['accuracy'])
    m2, _, _ = lm.MODELS[mt]()
    m2.compile(lm.Adam(lr=1e-3), loss='categorical_crossentropy', metrics=['accuracy'])
    m1.fit_generator(gen(), len(batches), 1, verbose=0, callbacks=[Rec()])
    plain = [m2.train_on_batch([v, a], l)[0] for v, a, l in batches]
    assert losses == plain


@pytest.mark.gpu
def test_fit_generator_reads_results_one_step_late_in_keras_order(gpu_required):
    """fit_generator enqueues step k + 1 before it reads step k's loss (l3_step_results_enqueue / _wait): the values equal those of
    the synchronous reader step by step, every callback sees begin(k), end(k) in Keras' order, and the deferred slot API returns
    what l3_step_results returns."""
    from l3embedding_amd import model as lm
    mt, B = 'tiny_L3', 4
    rng = np.random.RandomState(11)
    batches = []
    for _ in range(6):
        vid = rng.randint(0, 256, size=(B, 224, 224, 3)).astype(np.uint8)
        aud = rng.randint(-32768, 32768, size=(B, 1, 48000)).astype(np.int16)
        lab0 = rng.randint(0, 2, B)
        batches.append((vid, aud, np.stack([lab0, 1 - lab0], 1).astype(np.int32)))
    # slot API against the synchronous reader, two steps in flight
    e1, e2 = _lib.Engine(mt, B, seed=4), _lib.Engine(mt, B, seed=4)
    e2.set_params(e1.get_params())
    sync, deferred = [], []
    for k, b in enumerate(batches):
        e1.upload_batch_raw(*b)
        e1.step_resident(1e-3)
        sync.append(e1.step_results())
        e2.upload_batch_raw(*b)
        e2.step_resident(1e-3)
        e2.results_enqueue(k & 1)
        if k > 0:
            deferred.append(e2.results_wait((k - 1) & 1))        # read while step k is queued behind it
    deferred.append(e2.results_wait((len(batches) - 1) & 1))
    assert deferred == sync
    e1.close()
    e2.close()
    events = []

    class Rec(object):
        batch_hooks_are_passive = True          # they only record `logs`: the pipelined order is allowed (model._batch_hooks_passive)

        def on_train_begin(self, logs): pass
        def on_train_end(self, logs): pass
        def on_epoch_begin(self, e, logs): events.append(('epoch', e))
        def on_epoch_end(self, e, logs): events.append(('epoch_end', e, round(logs['loss'], 6)))
        def on_batch_begin(self, b, logs): events.append(('begin', b))
        def on_batch_end(self, b, logs): events.append(('end', b, logs['loss']))

    def gen():
        while True:
            for v, a, l in batches:
                yield [v, a], l
    m1, _, _ = lm.MODELS[mt]()
    m1.compile(lm.Adam(lr=1e-3), loss='categorical_crossentropy', metrics=['accuracy'])
    m2, _, _ = lm.MODELS[mt]()
    m2.compile(lm.Adam(lr=1e-3), loss='categorical_crossentropy', metrics=['accuracy'])
    hist = m1.fit_generator(gen(), 3, 2, verbose=0, callbacks=[Rec()])
    plain = [m2.train_on_batch([v, a], l)[0] for v, a, l in batches]
    order = [ev[:2] for ev in events]
    assert order == [('epoch', 0), ('begin', 0), ('end', 0), ('begin', 1), ('end', 1), ('begin', 2), ('end', 2), ('epoch_end', 0),
                     ('epoch', 1), ('begin', 0), ('end', 0), ('begin', 1), ('end', 1), ('begin', 2), ('end', 2), ('epoch_end', 1)]
    assert [ev[2] for ev in events if ev[0] == 'end'] == plain
    assert abs(hist.history['loss'][1] - float(np.mean(plain[3:]))) < 1e-6

    # a callback whose batch hooks touch the model (not declared passive) gets Keras' order against the COMPUTATION too:
    # on_batch_begin(k) runs before step k is launched (a learning rate set there applies to step k), and the weights seen in
    # on_batch_end(k) are those after step k, not after step k + 1
    seen = []

    class Strict(object):
        model = None

        def set_model(self, m): self.model = m
        def on_train_begin(self, logs): pass
        def on_train_end(self, logs): pass
        def on_epoch_begin(self, e, logs): pass
        def on_epoch_end(self, e, logs): pass
        def on_batch_begin(self, b, logs): seen.append(('begin', b, self.model._engine.optimizer_steps()[0] if self.model._engine else 0))
        def on_batch_end(self, b, logs): seen.append(('end', b, self.model._engine.optimizer_steps()[0]))

    m3, _, _ = lm.MODELS[mt]()
    m3.compile(lm.Adam(lr=1e-3), loss='categorical_crossentropy', metrics=['accuracy'])
    m3.fit_generator(gen(), 3, 1, verbose=0, callbacks=[Strict()])
    assert seen == [('begin', 0, 0), ('end', 0, 1), ('begin', 1, 1), ('end', 1, 2), ('begin', 2, 2), ('end', 2, 3)], seen


@pytest.mark.gpu
def test_dp_world2_one_gpu_gloo(gpu_required, tmp_path):
    """Two ranks of the real engine + DataParallelTrainer on ONE GPU (gloo all-reduces the CUDA gradient
    buckets through the host): both ranks must end on bit-identical weights, equal to an in-process
    emulation that sums the two shards' gradient arenas by hand before Adam.  Round 6: the BatchNorm MOVING statistics too --
    each rank gathers both shards' batch statistics and applies the two replica updates in rank order (multi_gpu_model calls
    the template model once per replica, training_utils.py:141-157): bit-identical on both ranks, equal to the emulation and to
    the oracle's virtual-rank step, and the same validation logits for the same rows whichever rank evaluates them."""
    import subprocess
    import sys
    import torch
    from l3embedding_amd.training_utils import get_slice_bounds, _DevArray
    steps = 2
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', WORLD_SIZE='2')
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dp_worker.py')
    procs = [subprocess.Popen([sys.executable, worker, str(tmp_path), str(steps)], env=dict(env, RANK=str(r)))
             for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    z = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r)) for r in range(2)]
    names = [k for k in z[0].files if k not in ('losses', 'val_logits', 'bn_updates')]
    assert any('moving_mean' in k for k in names) and any('moving_variance' in k for k in names)
    assert all(np.array_equal(z[0][k], z[1][k]) for k in names)          # replicas stay identical: weights AND moving statistics
    assert np.array_equal(z[0]['val_logits'], z[1]['val_logits'])        # the same rows validate alike on either rank
    assert z[0]['bn_updates'].tolist() == [steps, 2 * steps] == z[1]['bn_updates'].tolist()
    # in-process emulation: two shard engines, gradient arenas summed by hand
    mt, GB = 'tiny_L3', 6
    v, a, l = o.synthetic_batch(GB, seed=31)
    engs, flats = [], []
    for r in range(2):
        lo, hi = get_slice_bounds(GB, 2, r)
        e = _lib.Engine(mt, hi - lo, seed=13, global_batch=GB)
        ptr, n = e.grad_arena()
        engs.append((e, lo, hi))
        flats.append(torch.as_tensor(_DevArray(ptr, n), device='cuda:0'))
    for _ in range(steps):
        for e, lo, hi in engs:
            e.upload_batch(v[lo:hi], a[lo:hi], l[lo:hi])
            e.step_forward(True)
            for b in range(1, e.bucket_count()):
                e.step_backward_bucket(b)
            e.sync()
        total = flats[0] + flats[1]
        for f in flats:
            f.copy_(total)
        # the moving-statistics exchange by hand: both shards' packed batch statistics, rank-major, into each engine's buffer
        packed = []
        for e, _, _ in engs:
            ptr, n = e.bn_stats_pack()
            e.sync()
            packed.append(torch.as_tensor(_DevArray(ptr, n), device='cuda:0').clone())
        both = torch.cat(packed)
        for e, _, _ in engs:
            torch.as_tensor(_DevArray(e.bn_stats_replicas(2), both.numel()), device='cuda:0').copy_(both)
        torch.cuda.synchronize()
        for e, _, _ in engs:
            e.step_update(1e-3, 1.0)
            e.sync()
    W = engs[0][0].get_params()
    for k in names:
        assert np.array_equal(z[0][k], W[k.replace('|', '/')]), k
    # ... and against the oracle's virtual replicas (float64; the shards really differ here: rows 0-2 and 3-5)
    P = o.init_params(mt, seed=13)
    e0 = _lib.Engine(mt, 3, seed=13)
    P.update(e0.get_params())
    e0.close()
    adam, bn = o.AdamState(), o.BNMovingState(zero_debias=True)
    for _ in range(steps):
        o.dp_train_step(mt, P, adam, bn, v, a, l, 1e-3, 2)
    # (the moving VARIANCES: a convolution bias in front of a BatchNorm has a zero true gradient, Adam moves it by +-lr on rounding
    # noise, and the batch means follow it -- the variances do not see it)
    worst = max(relerr(z[0][k], P[k.replace('|', '/')]) for k in names if 'moving_variance' in k)
    print('moving variances of the 2-rank run against the oracle\'s two virtual replicas: %.1e' % worst)
    assert worst < 2e-3
    for e, _, _ in engs:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize('tower', ['audio', 'vision'])
def test_tower_step_matches_oracle(gpu_required, tower):
    """l3_tower_step (SURVEY 8(d) config "audio tower only"): training-mode forward of one sub-network and
    the backward pass from the stand-in loss mean(tower output), against the oracle's tower functions."""
    from collections import OrderedDict
    mt, B = 'cnn_L3_melspec2', 2
    P = o.init_params(mt, seed=21)
    v, a, l = o.synthetic_batch(B, seed=22)
    f = o.forward(mt, P, v, a, True, np.float64)
    prefix, ops, out, caches = (('audio_model', f['spec']['audio'], f['a'], f['ca']) if tower == 'audio'
                                else ('vision_model', f['spec']['vision'], f['v'], f['cv']))
    G = OrderedDict()
    o._tower_backward(prefix, ops, np.full(out.shape, 1.0 / out.size), caches, True, G)
    eng = _lib.Engine(mt, B)
    eng.set_params(P)
    eng.upload_batch(v, a, l)
    eng.tower_step(tower, backward=True)
    eng.sync()
    h0 = eng.activation('h0').reshape(B, -1)
    nv = f['v'].shape[1]
    got = h0[:, nv:] if tower == 'audio' else h0[:, :nv]
    assert np.abs(got - out).max() < 1e-4 * max(1.0, np.abs(out).max())
    worst_max = worst_l2 = 0.0
    for name, shape, trainable in eng.param_table():
        if trainable and name.startswith(prefix + '/') and name in G and not name.endswith('/bias'):
            g = eng.get_grad(name, shape).astype(np.float64)
            worst_max = max(worst_max, float(np.abs(g - G[name]).max() / (np.abs(G[name]).max() + 1e-30)))
            worst_l2 = max(worst_l2, float(np.sqrt(((g - G[name]) ** 2).sum() / ((G[name] ** 2).sum() + 1e-300))))
    print('%s tower gradients vs float64 oracle: worst max-error / max %.2e, worst relative L2 %.2e' % (tower, worst_max, worst_l2))
    # batch 2 through 8 BatchNorm-backward stages and the max-pool decisions: single elements move by a few % of the
    # tensor's largest gradient (measured 3.2e-2), the tensor as a whole by < 1 % (layer by layer the kernels are exact to
    # 3e-6: tests/test_layer_parity_gpu.py)
    assert worst_max < 8e-2 and worst_l2 < 2e-2, (worst_max, worst_l2)
    eng.close()


# BASELINE.json configs[1] at ITS batch (only B = 2 was held against the oracle): tests/golden/audio_tower_b64.npz,
# one float64 step of the audio tower at 64 samples (make_tower_golden.py).  Measured on MI355X: front-end 7.3e-5 dB, output 8.0e-6 of its
# range, gradients 4.6e-3 of the tensor RMS / 7.9e-4 relative L2 / 2.0e-4 of the norm -- an order of magnitude tighter than the full model's
# batch-64 distances (the stand-in loss mean(output) has no arg-max of a softmax behind it); bounds = 3x the measured.
TOWER_B64 = (1.5e-2, 2.5e-3, 6e-4)
@pytest.mark.gpu
def test_audio_tower_step_matches_the_golden_at_batch_64(gpu_required):
    """`l3_tower_step('audio')` -- kapre front-end (audio_model.py:367-369; the factored DFT), audio tower in training mode
    (audio_model.py:370-437), backward from mean(tower output) -- at batch 64 against the committed float64 vector: the tower
    output (the 512-wide embedding-side activations), the BatchNorm batch statistics of every layer, a seeded sample of every
    gradient and its norm."""
    z = np.load(os.path.join(GOLDEN, 'audio_tower_b64.npz'))
    mod = _mod()
    mt, B = str(z['model_type']), int(z['batch'])
    P = mod.perturbed_params(mt, int(z['param_seed']))
    v, a, l = o.synthetic_batch(B, seed=int(z['data_seed']))
    eng = _lib.Engine(mt, B)
    eng.set_params(P)
    eng.upload_batch(v, a, l)
    eng.tower_step('audio', backward=True)
    eng.sync()
    fe = eng.activation('audio_model/frontend').astype(np.float64)
    d_fe = float(np.abs(fe[mod.sample_idx('frontend', fe.size, 4096)] - z['frontend_sample']).max())
    h0 = eng.activation('h0').reshape(B, -1)
    out = h0[:, h0.shape[1] - z['out'].shape[1]:]
    d_out = float(np.abs(out - z['out']).max() / np.abs(z['out']).max())
    bound = TOWER_B64
    worst = [0.0, 0.0, 0.0]
    rows = []
    for name, shape, trainable in eng.param_table():
        if not trainable or 'gsamp:' + name not in z.files or float(z['gnorm:' + name]) < 1e-9:
            continue
        if name.endswith('/bias') or int(np.prod(shape)) == 1:
            continue            # zero true gradient in front of a BatchNorm / one cancelling sum over every pixel
        g = eng.get_grad(name, shape).astype(np.float64)
        idx = mod.sample_idx(name, g.size)
        ref, gnorm = z['gsamp:' + name], float(z['gnorm:' + name])
        err, nerr = mod.grad_metrics(g, ref, gnorm, idx)
        l2 = float(np.sqrt(((g.ravel()[idx] - ref) ** 2).sum() / ((ref ** 2).sum() + 1e-300)))
        rows.append((name, err, l2, nerr))
        worst = [max(worst[0], err), max(worst[1], l2), max(worst[2], nerr)]
    print('audio tower, batch 64: front-end %.2e dB, output %.2e of its range; gradients worst sampled err/rms %.3e, sampled L2 %.3e, norm %.3e'
          % (d_fe, d_out, worst[0], worst[1], worst[2]))
    for r in sorted(rows, key=lambda r: -r[1])[:4]:
        print('      %-50s err/rms %.3e  L2 %.3e  norm %.3e' % r)
    assert d_fe < 5e-4 and d_out < 3e-5
    assert len(rows) >= 20
    assert worst[0] < bound[0] and worst[1] < bound[1] and worst[2] < bound[2], worst
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(8, 56, 56, 128), (2, 256, 199, 64), (4, 28, 28, 512), (3, 33, 25, 64)])
def test_conv_delta_filters_shift_exactly(gpu_required, shape):
    """Oracle-free structural check at layer-sized shapes (odd widths included): a filter that is the
    identity on one tap makes the 3x3 'same' convolution a zero-padded shift, forward, and the data
    gradient the opposite shift -- which exercises every tile / halo / tail path of the Winograd kernel.
    (F(2x2,3x3): G g G^T of a one-hot filter is exact in fp32 except for 1/4 factors, so the tolerance is tiny;
    F(4x4,3x3), c >= 64: G holds 1/6 and 1/24 and the transforms multiply by up to 8 -- its layer budget applies.)"""
    n, h, w, c = shape
    tol = 3e-5 if c >= 64 else 2e-6
    rng = np.random.RandomState(c + h)
    x = rng.randn(n, h, w, c).astype(np.float32)
    dy = rng.randn(n, h, w, c).astype(np.float32)
    for kh, kw in ((0, 0), (1, 1), (2, 1), (0, 2)):
        wt = np.zeros((3, 3, c, c), np.float32)
        wt[kh, kw] = np.eye(c, dtype=np.float32)
        y = _lib.op_conv2d_fwd(x, wt, np.zeros(c, np.float32), True)
        want = np.zeros_like(x)
        dh, dw = kh - 1, kw - 1                      # y[p] = x[p + (dh, dw)]
        ys, xs = slice(max(0, -dh), h - max(0, dh)), slice(max(0, -dw), w - max(0, dw))
        ys2, xs2 = slice(max(0, dh), h - max(0, -dh)), slice(max(0, dw), w - max(0, -dw))
        want[:, ys, xs] = x[:, ys2, xs2]
        assert np.abs(y - want).max() < tol * np.abs(x).max(), (kh, kw)
        dx, dwg, db = _lib.op_conv2d_bwd(x, wt, dy, True)
        wantdx = np.zeros_like(dy)
        wantdx[:, ys2, xs2] = dy[:, ys, xs]          # dx[q] = dy[q - (dh, dw)]
        assert np.abs(dx - wantdx).max() < tol * np.abs(dy).max(), (kh, kw)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_conv_random_geometries(gpu_required, dtype):
    """Seeded sweep over geometries that steer the dispatch through every conv path (Winograd with all three
    tile-block widths and ragged tile rows, the 9-tap and the generic weight gradient, the bf16 kernels,
    the small-channel fallbacks, tails in every dimension), against the oracle."""
    rng = np.random.RandomState(2024)
    cases = []
    for _ in range(14):
        n = int(rng.randint(1, 4))
        h, w = int(rng.randint(1, 40)), int(rng.randint(1, 70))
        ci = int(rng.choice([8, 16, 24, 64, 72, 128, 192]))
        co = int(rng.choice([64, 128, 192, 32, 48]))
        cases.append((n, h, w, ci, co))
    cases += [(2, 7, 33, 64, 64), (1, 4, 8, 64, 128), (5, 3, 4, 128, 64), (1, 31, 2, 64, 64)]
    for (n, h, w, ci, co) in cases:
        x = rng.randn(n, h, w, ci).astype(np.float32)
        wt = (rng.randn(3, 3, ci, co) / np.sqrt(9 * ci)).astype(np.float32)
        b = rng.randn(co).astype(np.float32)
        dy = rng.randn(n, h, w, co).astype(np.float32)
        x64, w64, b64, dy64 = (t.astype(np.float64) for t in (x, wt, b, dy))
        with o.mixed_precision(dtype):
            y_ref = o.conv2d_fwd(x64, w64, b64, 'same')
            dx_ref, dw_ref, db_ref = o.conv2d_bwd(x64, w64, dy64, 'same')
        y = _lib.op_conv2d_fwd(x, wt, b, True, dtype=dtype)
        dx, dw, db = _lib.op_conv2d_bwd(x, wt, dy, True, dtype=dtype)
        tag = (n, h, w, ci, co)
        # fp32: Winograd F(4x4,3x3) where the (forward / data-gradient) conv has >= 64 input channels -- its own budget
        f4y = dtype == 'f32' and ci >= 64 and ci % 8 == 0 and co % 64 == 0
        f4x = dtype == 'f32' and co >= 64 and co % 8 == 0 and ci % 64 == 0
        assert relerr(y, y_ref) < (3e-5 if f4y else 5e-6), tag
        assert relerr(dx, dx_ref) < (3e-5 if f4x else 5e-6) and relerr(dw, dw_ref) < 5e-6 and relerr(db, db_ref) < 5e-6, tag


@pytest.mark.gpu
def test_fp32_step_runs_the_winograd_kernels(gpu_required):
    """No silent fallback: in an fp32 engine all three convolution families of cnn_L3_melspec2 must take their
    Winograd kernels -- the engine's own account of the MFMA flops it ISSUES is, for the weight gradient, 16/36 of the
    direct-convolution count (F(3x3,2x2) on the 14 3x3 layers, plus tile padding, plus the direct first-layer launches),
    and for forward / data gradient lower still: F(4x4,3x3) (9/36, plus tile padding) on all 14 -- about 0.26 of direct.
    Nowhere near 1."""
    mt, B = 'cnn_L3_melspec2', 2
    v, a, l = o.synthetic_batch(B, seed=1)
    eng = _lib.Engine(mt, B, seed=0)
    eng.upload_batch(v, a, l)
    eng.step_resident(1e-4)
    eng.profile_enable(True)
    eng.step_resident(1e-4)
    eng.sync()
    pr = eng.profile_read()
    eng.close()
    for fam in ('conv_fwd', 'conv_dgrad', 'conv_wgrad'):
        ratio = pr[fam]['executed_flops'] / pr[fam]['flops']
        print('%s: issued / direct flops = %.3f' % (fam, ratio))
        lo, hi = (0.44, 0.56) if fam == 'conv_wgrad' else (0.24, 0.32)
        assert lo < ratio < hi, (fam, ratio)


@pytest.mark.parametrize('ncu', ['256', '24'])
def test_split_tail_of_the_f4_kernel_in_a_full_step(gpu_required, monkeypatch, ncu):
    """conv_wino4_launch, the tail: when a layer's tile blocks do not divide by the CU count, the last, partial round goes to a second
    launch that slices every tail block over its input channels, and wino4_tail_reduce_kernel sums the slices, adds the bias and
    takes the BatchNorm statistics (forward) / the fused BatchNorm-backward reduction (data gradient).  Product engines only do
    this for a tower running on its own (ConvGeom::solo); here it is forced onto the two-tower step (L3_W4_TAIL=2), once with
    the real CU count (every small-batch layer is ALL tail) and once on an emulated 24-CU chip (full rounds + tail): same sums in
    another order -- loss and every gradient must agree with the one-pass engine (L3_W4_TAIL=0) to round-off."""
    mt, B = 'cnn_L3_melspec2', 4
    v, a, l = o.synthetic_batch(B, seed=21)
    monkeypatch.setenv('L3_W4_NCU', ncu)
    res = {}
    for tail in ('0', '2'):
        monkeypatch.setenv('L3_W4_TAIL', tail)
        eng = _lib.Engine(mt, B, seed=6)
        loss, _ = eng.train_step(v, a, l, 1e-4)
        res[tail] = (loss, eng.get_grads())
        eng.close()
    assert abs(res['0'][0] - res['2'][0]) < 1e-5 * max(1.0, abs(res['0'][0]))
    dist = {}
    for name, g0 in res['0'][1].items():
        if (name.endswith('/bias') and not name.startswith('dense')) or g0.size == 1:
            continue        # zero up to round-off: a BatchNorm follows
        g0 = g0.astype(np.float64)
        dist[name] = float(np.sqrt(((res['2'][1][name] - g0) ** 2).sum() / ((g0 ** 2).sum() + 1e-300)))
    top = sorted(dist.items(), key=lambda kv: -kv[1])[:4]
    print('split tails (%s CUs) vs one pass, largest relative L2 gradient distances: %s' % (ncu, ', '.join('%s %.1e' % kv for kv in top)))
    # Every convolution output moves in its last bits (another summation order over the input channels); besides round-off that
    # flips the ReLU mask of the odd element sitting at zero, which moves ONE term of a per-channel sum (measured here: one flip
    # under vision batch_normalization_8 -- its beta and conv2d_7's kernel move by 2-3 % of the tensor's maximum in that channel,
    # scripts/probes/dbg_tail.py) -- hence a distance over the whole tensor; a wrong slice, bias or mask is O(1) in it.
    for name, d in dist.items():
        assert d < 2e-2, (name, d)


@pytest.mark.parametrize('algo,ncu,B', [('f4x4', '256', 8), ('f4x4', '24', 4), ('f4x4', '8', 2), ('f2x2', '256', 8), ('f2x2', '24', 4)])
def test_dynamic_tile_block_assignment_is_bit_identical(gpu_required, monkeypatch, algo, ncu, B):
    """Persistent convolution grids under a data-parallel step hand their tile blocks out through work counters (device_common.h
    wq_*: eight per-XCD queues, a workgroup claims the block after next one block ahead, an empty queue steals from the others,
    the last workgroup to leave resets the counters) instead of the static stride b, b + grid, ...: a collective that holds CUs
    then costs the late workgroups' share only (training_utils.py:141-170 is what the collectives stand in for).
    Which workgroup computes a tile block must not enter the arithmetic: three training steps with the counters forced on
    (L3_W4_DYNAMIC=1) and off end on bit-identical weights, statistics and gradients -- on the real chip, on an emulated 24-CU
    chip (many blocks per workgroup, ragged queues) and on 8 CUs (one workgroup per queue: every queue runs dry and steals)."""
    mt = 'cnn_L3_melspec2'
    v, a, l = o.synthetic_batch(B, seed=33)
    monkeypatch.setenv('L3_W4_NCU', ncu)
    res = {}
    for dyn in ('0', '1'):
        monkeypatch.setenv('L3_W4_DYNAMIC', dyn)
        eng = _lib.Engine(mt, B, seed=6, fp32_conv=algo)
        losses = [eng.train_step(v, a, l, 1e-4)[0] for _ in range(3)]
        res[dyn] = (losses, eng.get_params(), eng.get_grads())
        eng.close()
    assert res['0'][0] == res['1'][0]
    for k in (1, 2):
        bad = [n for n in res['0'][k] if not np.array_equal(res['0'][k][n], res['1'][k][n])]
        assert bad == [], bad[:5]


def test_split_tail_step_matches_golden(gpu_required, monkeypatch):
    """... and the forced-split two-tower step against the float64 golden of batch 8, same bounds as the one-pass engine."""
    monkeypatch.setenv('L3_W4_TAIL', '2')
    monkeypatch.setenv('L3_W4_NCU', '40')
    test_training_step_matches_golden(gpu_required, 'cnn_L3_melspec2_b8.npz')


def test_tower_models_predict_and_merge(gpu_required):
    """construct_cnn_L3_melspec2_audio_model() / construct_cnn_L3_orig_inputbn_vision_model() (audio_model.py:335-442,
    vision_model.py:102-195) as free-standing models: predict() = the tower's flattened output against the oracle, and
    L3_merge_audio_vision_models (model.py:7-35) carries the towers' weights into the AVC model."""
    mt, B = 'cnn_L3_melspec2', 2
    mod = _mod()
    P = mod.perturbed_params(mt, 77)
    v, a, l = o.synthetic_batch(B, seed=78)
    ref = o.forward(mt, P, v, a, False, np.float64)
    am, x_a, _ = model.construct_cnn_L3_melspec2_audio_model()
    vm, x_i, _ = model.construct_cnn_L3_orig_inputbn_vision_model()
    for tower in (am, vm):
        names = tower._names()
        tower.set_weights([P[n] for n in names])
    ya, yv = am.predict(a), vm.predict(v)
    assert ya.shape == (B, 512) and yv.shape == (B, 512)
    assert np.abs(ya - ref['a']).max() < 2e-4 * max(1.0, np.abs(ref['a']).max())
    assert np.abs(yv - ref['v']).max() < 2e-4 * max(1.0, np.abs(ref['v']).max())
    m, _, _ = model.L3_merge_audio_vision_models(vm, x_i, am, x_a, 'cnn_L3_melspec2')
    head = OrderedDict((n, P[n]) for n in P if n.startswith('dense_'))
    m._assign(head)
    logits = m.predict_logits([v, a])
    assert np.abs(logits - ref['logits']).max() < LOGIT_TOL
    for mm in (am._parent, vm._parent, m):
        mm._drop_engines()


def test_fp32_conv_algorithm_is_configuration(gpu_required, monkeypatch):
    """l3_config.fp32_conv (include/l3hip.h L3_FP32_CONV_*): a caller that wants the tighter parity of F(2x2,3x3) selects it
    through the boundary -- no debug knob involved (L3_DEBUG_KNOBS unset here).  The engine's account of issued flops shows
    which kernels ran (forward / data gradient: 16/36 of direct + tile padding against 9/36), and on a layer-sized problem
    the F(2x2,3x3) engine lands closer to the float64 oracle."""
    monkeypatch.delenv('L3_DEBUG_KNOBS', raising=False)
    mt, B = 'cnn_L3_melspec2', 2
    mod = _mod()
    P = mod.perturbed_params(mt, 101)
    v, a, l = o.synthetic_batch(B, seed=202)
    z = np.load(os.path.join(GOLDEN, 'cnn_L3_melspec2_b2.npz'))
    ratios, dist = {}, {}
    exp = _lib.experiments_built()          # the split-bf16 experiment exists only in an L3_BUILD_EXPERIMENTS=1 library
    for algo in ('f4x4', 'f2x2') + (('f2x2_bf16x6',) if exp else ()):
        eng = _lib.Engine(mt, B, seed=0, fp32_conv=algo)
        eng.set_params(P)
        _, logits = eng.forward(v, a, training=True)
        dist[algo] = float(np.abs(logits - z['train_logits']).max())
        eng.upload_batch(v, a, l)
        eng.step_resident(1e-4)
        eng.profile_enable(True)
        eng.step_resident(1e-4)
        eng.sync()
        pr = eng.profile_read()
        eng.close()
        ratios[algo] = {f: pr[f]['executed_flops'] / pr[f]['flops'] for f in ('conv_fwd', 'conv_dgrad', 'conv_wgrad')}
    print('issued / direct flops:', ratios, ' |logits - float64|:', dist)
    for fam in ('conv_fwd', 'conv_dgrad'):
        assert 0.24 < ratios['f4x4'][fam] < 0.32 and 0.44 < ratios['f2x2'][fam] < 0.56, ratios
        # split-bf16 F(2x2,3x3): six bf16 products per fp32 one -- 6 x 16/36 of direct (+ tile padding), counted as bf16 flops
        assert not exp or 2.6 < ratios['f2x2_bf16x6'][fam] < 3.6, ratios
    assert abs(ratios['f4x4']['conv_wgrad'] - ratios['f2x2']['conv_wgrad']) < 1e-6          # the weight gradient is F(3x3,2x2) in all
    assert dist['f2x2'] < LOGIT_TOL and dist['f4x4'] < LOGIT_TOL
    if exp:
        assert abs(ratios['f4x4']['conv_wgrad'] - ratios['f2x2_bf16x6']['conv_wgrad']) < 1e-6
        assert dist['f2x2_bf16x6'] < LOGIT_TOL
        assert dist['f2x2_bf16x6'] < 2 * dist['f2x2'] + 1e-5          # fp32-grade: as close to float64 as the fp32 F(2x2,3x3) engine
    else:
        with pytest.raises(_lib.L3Error, match='L3_BUILD_EXPERIMENTS'):      # not a product configuration: refused, not silently replaced
            _lib.Engine(mt, B, fp32_conv='f2x2_bf16x6')
    with pytest.raises(ValueError):
        _lib.Engine(mt, B, fp32_conv='direct')
    cfg = _lib.L3Config()
    cfg.struct_size = ctypes.sizeof(_lib.L3Config)
    cfg.model_type, cfg.batch, cfg.fp32_conv = 4, 1, 7
    h = ctypes.c_void_p()
    assert _lib.load().l3_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == -1                  # L3_EINVAL


def test_integration_md_stub_runs_as_written(gpu_required, tmp_path):
    """INTEGRATION.md section 2 is the binding a maintainer would paste: run that very block (batch cut to 2)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    block = re.search(r"## 2\. Minimal ctypes stub.*?```python\n(.*?)```", text, re.S).group(1)
    block = block.replace('(64, ', '(2, ').replace("b'cnn_L3_melspec2'), 64,", "b'cnn_L3_melspec2'), 2,")
    block += "\nprint('STUB', rc, loss.value, acc.value)\n"
    res = subprocess.run([sys.executable, '-c', block], cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'STUB 0' in res.stdout, res.stdout + res.stderr
    loss = float(res.stdout.split('STUB 0')[1].split()[0])
    assert np.isfinite(loss) and 0.3 < loss < 3.0


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_bn_backward_partials_from_the_dgrad_epilogue(gpu_required, monkeypatch, dtype):
    """The BatchNorm backward reduction (sum of the masked gradient, sum of masked gradient * x_hat) of the first
    BatchNorm of every block is taken in the epilogue of the data gradient that produces its dL/dy (kernels.h
    BnBwdFuse: the Winograd kernel in fp32 engines, the LDS-halo kernel on the bf16-stored tensors of mixed-precision
    engines) instead of a pass over x and dy.  Same sums in another order: every gradient of a training
    step must agree with the unfused engine (L3_BNBWD_FUSE=0) to fp32 round-off, and with the float64 oracle as before
    (the golden tests run the fused form)."""
    mt, B = 'cnn_L3_melspec2', 4
    v, a, l = o.synthetic_batch(B, seed=11)
    grads = {}
    for fuse in ('1', '0'):
        monkeypatch.setenv('L3_BNBWD_FUSE', fuse)
        eng = _lib.Engine(mt, B, seed=5, dtype=dtype)
        loss, _ = eng.train_step(v, a, l, 1e-4)
        grads[fuse] = (loss, eng.get_grads())
        eng.close()
    assert abs(grads['1'][0] - grads['0'][0]) < 1e-6 * max(1.0, abs(grads['0'][0]))
    worst = 0.0
    per_bn = {}
    for name, g0 in grads['0'][1].items():
        g1 = grads['1'][1][name]
        if (name.endswith('/bias') and not name.startswith('dense')) or g0.size == 1:
            continue        # zero up to round-off: a BatchNorm follows (biases; the single-channel input BatchNorm's gamma / beta)
        d = float(np.abs(g1 - g0).max() / (np.abs(g0).max() + 1e-30))
        worst = max(worst, d)
        # round-off of a different summation order, amplified by the BatchNorm stages below the layer (cf. the fp32
        # NumPy oracle's own distance to float64 in profiles/r02_parity_distances.txt); a wrong mask or x_hat is O(1)
        # fp32 measured 9e-6.  Mixed precision: a sum that differs in its last bits can move a bf16-stored gradient element
        # below it to the neighbouring bfloat16 (2^-8 relative), which the BatchNorm stages further down carry on and amplify
        # (measured: 1e-7 at the deepest fused layer, growing to 1.5e-2 at single parameters near the input)
        assert d < (2e-4 if dtype == 'f32' else 6e-2), (name, d)
        if 'batch_normalization' in name:
            per_bn.setdefault(name.split('/')[0], {})[int(name.split('/')[1].rsplit('_', 1)[1])] = max(
                d, per_bn.get(name.split('/')[0], {}).get(int(name.split('/')[1].rsplit('_', 1)[1]), 0.0))
    for tower, d_by_layer in per_bn.items():
        # the last block of a tower sees identical inputs in both engines: its two BatchNorms (the second one is not fused,
        # the first one is the deepest fused layer) must agree to summation-order round-off
        for layer in sorted(d_by_layer)[-2:]:
            assert d_by_layer[layer] < 1e-5, (tower, layer, d_by_layer[layer])
    print('fused vs unfused BatchNorm-backward reduction (%s): worst gradient distance / max = %.2e' % (dtype, worst))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_first_layer_weight_gradient_forms_its_own_output_gradient(gpu_required, dtype, monkeypatch):
    """The BatchNorm behind a tower's first convolution does not write the convolution's output gradient (the largest tensor of
    the network; nothing but the first layer's weight gradient reads it): bn_bwd_fast leaves its backward coefficients and
    conv_first_wgrad_kernel<CA, FUSE> forms dY = cA mask(dA) + cB y + cC per element as it streams (kernels.h FirstWgFuse) -- the
    expression bn_bwd_apply_fast_kernel evaluates (up to the compiler's choice of fused multiply-adds: measured 3e-7 of the
    tensor): against the engine that writes dY (L3_FIRST_WG_FUSE=0) the first layer's kernel gradient and the input BatchNorm's
    gamma / beta agree to 1e-5 of the tensor's maximum, every other gradient is bit-identical (nothing else changes), and the
    first convolution's bias gradient -- now the ones-channel row of the centre tap instead of the column sums of dY -- equals
    the old one to summation round-off of a quantity that is itself zero up to round-off (a BatchNorm follows)."""
    mt, B = 'cnn_L3_melspec2', 3
    v, a, l = o.synthetic_batch(B, seed=13)
    grads = {}
    for fuse in ('1', '0'):
        monkeypatch.setenv('L3_FIRST_WG_FUSE', fuse)
        eng = _lib.Engine(mt, B, seed=7, dtype=dtype)
        loss, _ = eng.train_step(v, a, l, 1e-4)
        grads[fuse] = (loss, eng.get_grads())
        eng.close()
    assert grads['1'][0] == grads['0'][0]
    first_bias = ('vision_model/conv2d_1/bias', 'audio_model/conv2d_8/bias')
    for name, g0 in grads['0'][1].items():
        g1 = grads['1'][1][name]
        if name in first_bias:
            scale = max(float(np.abs(grads['0'][1][name.replace('/bias', '/kernel')]).max()), 1e-30)
            assert float(np.abs(g1 - g0).max()) < 1e-3 * scale, (name, float(np.abs(g1 - g0).max()), scale)
        elif name.rsplit('/', 1)[0] in ('vision_model/conv2d_1', 'audio_model/conv2d_8', 'vision_model/batch_normalization_1',
                                        'audio_model/batch_normalization_10'):
            if g0.size == 1:
                continue        # the single-channel input BatchNorm's gamma / beta: sums that cancel to round-off (as the tests above)
            assert float(np.abs(g1 - g0).max()) <= 1e-5 * float(np.abs(g0).max()), (name, float(np.abs(g1 - g0).max()))
        else:
            assert np.array_equal(g1, g0), (name, float(np.abs(g1 - g0).max()))
    assert all(n in grads['0'][1] for n in first_bias)


@pytest.mark.gpu
@pytest.mark.parametrize('kernel', ['mfma16', 'generic'])
def test_first_layer_weight_gradient(gpu_required, kernel, monkeypatch):
    """Weight gradient of a tower's first convolution in the form the engine computes it (2 or 4 input channels:
    the normalised input + the ones channel that carries the input BatchNorm's beta; 64 filters) -- the dY-streaming
    v_mfma_f32_16x16x4_f32 kernel of conv_first.hip against the float64 oracle, widths that are not multiples of the
    4-pixel unit and fewer units than waves included; 'generic' = the kernel it replaces."""
    if kernel == 'generic':
        monkeypatch.setenv('L3_FIRST_WGRAD', '0')
    rng = np.random.RandomState(5)
    for (n, h, w, ci) in [(2, 64, 50, 2), (2, 56, 56, 4), (1, 9, 7, 4), (3, 5, 199, 2), (1, 1, 1, 2), (1, 3, 2, 4)]:
        x = rng.randn(n, h, w, ci).astype(np.float32)
        x[..., -1] = 1.0
        dy = (rng.randn(n, h, w, 64) * 1e-3).astype(np.float32)
        wt = np.zeros((3, 3, ci, 64), np.float32)
        _, dw_ref, _ = o.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64), 'same')
        _, dw, _ = _lib.op_conv2d_bwd(x, wt, dy, True)
        assert relerr(dw, dw_ref) < 3e-6, ((n, h, w, ci), kernel, relerr(dw, dw_ref))


@pytest.mark.gpu
@pytest.mark.parametrize('uc', ['auto', '8', '4', '2', 'direct'])
def test_wgrad_winograd_unit_shapes(gpu_required, uc, monkeypatch):
    """fp32 weight gradient as Winograd F(3x3, 2x2) (conv_wgrad_wino.hip) with each of its 8-tile unit shapes
    (1x8, 2x4, 4x2) forced on geometries with odd heights / widths, partial units, one and many split-K slices,
    against the float64 oracle; 'direct' = the same cases through the 9-tap kernel it replaces."""
    if uc == 'direct':
        monkeypatch.setenv('L3_WG_WINO', '0')
    elif uc != 'auto':
        monkeypatch.setenv('L3_WGW_UC', uc)
    rng = np.random.RandomState(77)
    worst = 0.0
    for (n, h, w, ci, co) in [(2, 64, 49, 128, 64), (1, 33, 25, 64, 128), (3, 7, 5, 64, 64), (1, 1, 1, 64, 64),
                              (2, 32, 24, 256, 192), (1, 128, 99, 64, 64), (5, 2, 17, 64, 64)]:
        x = np.maximum(rng.randn(n, h, w, ci), 0).astype(np.float32)
        dy = (rng.randn(n, h, w, co) * 1e-3).astype(np.float32)
        wt = np.zeros((3, 3, ci, co), np.float32)
        _, dw_ref, _ = o.conv2d_bwd(x.astype(np.float64), wt.astype(np.float64), dy.astype(np.float64), 'same')
        _, dw, _ = _lib.op_conv2d_bwd(x, wt, dy, True)
        err = relerr(dw, dw_ref)
        worst = max(worst, err)
        assert err < 3e-6, ((n, h, w, ci, co), uc, err)
    print('wgrad %s: worst max-error / max vs float64 = %.2e' % (uc, worst))


@pytest.mark.gpu
@pytest.mark.parametrize('variant', ['halo_auto', 'halo_pw32', 'halo_pw16', 'halo_flat', 'halo_flat5', 'halo_2d', 'halo_regfilter', 'tap_tiles', 'wgrad_cvt',
                                     'wgrad_tapsplit', 'wgrad_8x8', 'wgrad_4x16'])
def test_conv_bf16_stored_random_geometries(gpu_required, variant, monkeypatch):
    """Stored-operand mixed-precision convolution (the form an L3_DTYPE_BF16 engine runs) over geometries that are
    ragged against every tile shape: the LDS-halo kernel with 8x32 and 16x16 patches and with flat tiles of 256 consecutive
    pixels (conv_bf16_halo.hip, Cout a multiple of 128; halo_flat forces the flat tiles wherever their halo fits, also where
    a 2-D patch would not pad: tiles that start mid-row, span several whole images -- 5 x 3 x 4 is ONE tile with four zero
    rows inside --, end short of 256 pixels) and the tap-by-tap kernel (conv_bf16.hip), forward and data gradient, against
    the oracle."""
    if variant in ('halo_flat5', 'wgrad_tapsplit', 'halo_regfilter'):
        need_experiments()
    if variant == 'halo_regfilter':          # 64-channel blocks with the filter in registers (measured slower: profiles/r06_halo64_regfilter.txt)
        monkeypatch.setenv('L3_HALO_BREG', '1')
    monkeypatch.setenv('L3_BF16_HALO', '0' if variant == 'tap_tiles' else '1')
    monkeypatch.setenv('L3_WG_TR', '0' if variant == 'wgrad_cvt' else '1')       # transpose-read vs convert-in-register wgrad
    monkeypatch.setenv('L3_WG_TR_TS', '2' if variant == 'wgrad_tapsplit' else '1')   # 8 waves with the taps split (measured, not the default)
    if variant in ('wgrad_8x8', 'wgrad_4x16'):                                       # patch shape of the transpose-read weight gradient
        monkeypatch.setenv('L3_WG_TR_SQ', '1' if variant == 'wgrad_8x8' else '0')    # (default: whichever pads the image less)
    if variant.startswith('halo_pw'):
        monkeypatch.setenv('L3_HALO_PW', variant[-2:])
    if variant in ('halo_flat', 'halo_flat5', 'halo_2d'):
        monkeypatch.setenv('L3_HALO_FLAT', '0' if variant == 'halo_2d' else '2')
    if variant == 'halo_flat5':              # the double-buffered, swizzled form of the flat tiles (measured, not the default)
        monkeypatch.setenv('L3_HALO_FLAT_MODE', '5')
    rng = np.random.RandomState(77)
    cases = [(1, 8, 32, 64, 128), (2, 9, 33, 64, 128), (1, 17, 15, 128, 256), (3, 5, 50, 64, 128), (2, 31, 7, 192, 128),
             (1, 40, 70, 64, 128), (2, 1, 1, 64, 128), (1, 16, 16, 128, 128), (2, 9, 33, 64, 64), (1, 20, 45, 128, 64),
             (1, 33, 17, 192, 64)]
    if variant in ('halo_flat', 'halo_flat5'):
        cases += [(5, 3, 4, 64, 128), (3, 28, 28, 64, 128), (2, 56, 56, 64, 128), (3, 32, 24, 128, 128), (2, 64, 49, 64, 256),
                  (7, 6, 5, 64, 128), (1, 13, 56, 192, 128)]
    for (n, h, w, ci, co) in cases:
        x = np.maximum(rng.randn(n, h, w, ci), 0).astype(np.float32)
        wt = (rng.randn(3, 3, ci, co) / np.sqrt(9 * ci)).astype(np.float32)
        b = rng.randn(co).astype(np.float32)
        dy = rng.randn(n, h, w, co).astype(np.float32)
        x64, w64, b64, dy64 = (t.astype(np.float64) for t in (x, wt, b, dy))
        with o.mixed_precision('bf16'):
            y_ref = o.conv2d_fwd(x64, w64, b64, 'same')
            dx_ref, dw_ref, db_ref = o.conv2d_bwd(x64, w64, dy64, 'same')
        y = _lib.op_conv2d_fwd(x, wt, b, True, dtype='bf16_stored')
        # the data gradient of an (ci -> co) layer is a (co -> ci) convolution: swap the roles so that IT has Cout % 128 == 0
        wt2 = (rng.randn(3, 3, co, ci) / np.sqrt(9 * co)).astype(np.float32)
        x2 = rng.randn(n, h, w, co).astype(np.float32)
        dy2 = rng.randn(n, h, w, ci).astype(np.float32)
        with o.mixed_precision('bf16'):
            dx2_ref, _, _ = o.conv2d_bwd(x2.astype(np.float64), wt2.astype(np.float64), dy2.astype(np.float64), 'same')
        dx2, _, _ = _lib.op_conv2d_bwd(x2, wt2, dy2, True, dtype='bf16_stored')
        dx, dw, db = _lib.op_conv2d_bwd(x, wt, dy, True, dtype='bf16_stored')
        tag = (variant, n, h, w, ci, co)
        assert relerr(y, y_ref) < 5e-6, tag
        assert relerr(dx2, dx2_ref) < 5e-6, tag
        assert relerr(dx, dx_ref) < 5e-6 and relerr(dw, dw_ref) < 5e-6 and relerr(db, db_ref) < 5e-6, tag


_HOST_WAIT_WORKER = r'''
import os, sys, time, json
import numpy as np
sys.path.insert(0, sys.argv[1])
from l3embedding_amd import _lib
from oracle import l3_oracle as o
B = 8
v, a, l = o.synthetic_batch(B, seed=5)
e = _lib.Engine('cnn_L3_melspec2', B, seed=3)
e.upload_batch(v, a, l)
for _ in range(2):
    e.step_resident(1e-4)
e.sync()
w0, c0 = time.perf_counter(), time.thread_time()
for _ in range(12):
    e.step_resident(1e-4)
e.sync()
w1, c1 = time.perf_counter(), time.thread_time()
loss, acc = e.step_results()
print(json.dumps({'loss': float(loss), 'wall': w1 - w0, 'thread_cpu': c1 - c0}))
'''


@pytest.mark.gpu
def test_library_host_waits_do_not_spin(gpu_required):
    """The library waits for the GPU by sleep-polling hipStreamQuery (csrc/knobs.h stream_wait): the calling thread
    takes next to no CPU while a queue of training steps runs, L3_HOST_WAIT=spin is hipStreamSynchronize, and the arithmetic does
    not depend on which of the two waited."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    res = {}
    for mode in ('poll', 'spin'):
        env = dict(os.environ)
        env.pop('L3_HOST_WAIT', None)
        if mode == 'spin':
            env['L3_HOST_WAIT'] = 'spin'
        out = subprocess.run([sys.executable, '-c', _HOST_WAIT_WORKER, root], env=env, stdout=subprocess.PIPE, timeout=300, check=True)
        res[mode] = json.loads(out.stdout.decode().strip().splitlines()[-1])
    print(res)
    assert res['poll']['loss'] == res['spin']['loss']
    assert res['poll']['wall'] > 0.02                                  # there was something to wait for
    assert res['poll']['thread_cpu'] < 0.35 * res['poll']['wall'], res
    assert res['poll']['wall'] < 1.15 * res['spin']['wall'] + 0.01, res     # ... and sleeping costs no time
