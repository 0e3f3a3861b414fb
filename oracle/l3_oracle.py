"""CPU oracle for the L3-Net AVC training path  --  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the arithmetic that the reference
(marl/l3embedding) *causes* Keras 2.0.9 / TensorFlow 1.4 / kapre 0.1.x to run for
one AVC training step.  It is the checker for the HIP path in `l3embedding_amd/`.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it; the product never does.

PARITY UNPINNED.  The reference holds no golden vectors, known-answer tests or
fixtures for this path and its third-party runtime
(keras, tensorflow, kapre, librosa) is not installable here, so this oracle cannot
be pinned against reference outputs.  It is pinned instead against
  (1) the reference's structural fixtures (parameter counts / output shapes in
      notebooks/test_load_converted_model.ipynb:100-214),
  (2) an independent PyTorch-CPU autograd build of the same graph
      (tests/torch_ref.py), and
  (3) finite-difference gradient checks.
Third-party semantics restated from the pinned versions are marked [3P].

Reference call sites followed (paths relative to /root/reference):
  l3embedding/audio_model.py:8-115,118-223,225-332,335-442,490-541   audio towers
  l3embedding/vision_model.py:7-99,102-195,221-265                    vision towers
  l3embedding/model.py:7-35,198-313                                   merge + head + registry
  l3embedding/train.py:186,189,269-284                                preprocessing, loss, Adam
  l3embedding/audio.py:4-31                                           pcm2float
  l3embedding/training_utils.py:121-170                               DP batch slicing, per-replica model calls

Layout conventions: activations NHWC, conv kernels HWIO, dense kernels (in, out).
"""
from collections import OrderedDict

import numpy as np

BN_EPS = 1e-3          # [3P] keras BatchNormalization default epsilon
BN_MOMENTUM = 0.99     # [3P] keras BatchNormalization default momentum
L2_WEIGHT = 1e-5       # audio_model.py:351, vision_model.py:118, model.py:24
K_EPSILON = 1e-7       # [3P] keras.backend.epsilon()
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-8   # [3P] keras.optimizers.Adam defaults


# ----------------------------------------------------------------------------
# A6: input preprocessing (train.py:186,189; audio.py:4-31)
# ----------------------------------------------------------------------------
def pcm2float(sig, dtype=np.float32):
    """audio.py:4-31 -- (sig - offset) / abs_max with abs_max = 2**(bits-1)."""
    sig = np.asarray(sig)
    if sig.dtype.kind not in 'iu':
        raise TypeError("'sig' must be an array of integers")
    dtype = np.dtype(dtype)
    if dtype.kind != 'f':
        raise TypeError("'dtype' must be a floating point type")
    i = np.iinfo(sig.dtype)
    abs_max = 2 ** (i.bits - 1)
    offset = i.min + abs_max
    return (sig.astype(dtype) - offset) / abs_max


def preprocess_video(u8):
    """train.py:186 -- 2 * img_as_float(uint8).astype('float32') - 1.

    [3P] skimage.img_as_float(uint8) = x / 255 in float64; the cast to float32
    happens before the affine map, which numpy then evaluates in float32.
    """
    u8 = np.asarray(u8)
    assert u8.dtype == np.uint8
    f = (u8.astype(np.float64) / 255.0).astype(np.float32)
    return (2 * f - 1).astype(np.float32)


# ----------------------------------------------------------------------------
# A2 front-end: kapre Spectrogram / Melspectrogram  [3P]
# ----------------------------------------------------------------------------
def stft_kernels(n_dft):
    """[3P] kapre.backend.get_stft_kernels: Hann(periodic)-windowed cos / -sin
    DFT bases, shape (n_dft, n_dft//2+1), float32 like K.floatx()."""
    nb = n_dft // 2 + 1
    t = np.arange(n_dft, dtype=np.float64)
    w_ks = np.arange(nb, dtype=np.float64) * 2 * np.pi / float(n_dft)
    real = np.cos(w_ks.reshape(-1, 1) * t.reshape(1, -1))
    imag = -np.sin(w_ks.reshape(-1, 1) * t.reshape(1, -1))
    # scipy-style hann(M, sym=False): 0.5 - 0.5 cos(2 pi n / M)
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * t / n_dft)).astype(np.float32)
    real = (real * win.reshape(1, -1)).T
    imag = (imag * win.reshape(1, -1)).T
    return real.astype(np.float32), imag.astype(np.float32)


def _hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def _mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_basis(sr, n_fft, n_mels, fmin=0.0, fmax=None, htk=True, norm=1):
    """[3P] librosa 0.5.1 filters.mel (requirements.txt:1) as used by
    kapre.backend.mel.  Returns (n_mels, 1 + n_fft//2) float64.  Only htk=True is
    needed: audio_model.py:258,368."""
    assert htk, "reference only uses htk=True"
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)))
    fftfreqs = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    mels = np.linspace(_hz_to_mel_htk(fmin), _hz_to_mel_htk(fmax), n_mels + 2)
    mel_f = _mel_to_hz_htk(mels)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    if norm == 1:
        enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
        weights *= enorm[:, np.newaxis]
    return weights


def frontend_constants(kind):
    """kapre layer weights for front-end `kind`, keyed like kapre's variables."""
    cfg = FRONTENDS[kind]
    real, imag = stft_kernels(cfg['n_dft'])
    out = OrderedDict(real_kernels=real.reshape(cfg['n_dft'], 1, 1, -1),
                      imag_kernels=imag.reshape(cfg['n_dft'], 1, 1, -1))
    if cfg['n_mels']:
        fb = mel_basis(48000, cfg['n_dft'], cfg['n_mels'], 0.0, None, True, 1)
        out['freq2mel'] = fb.T.astype(np.float32)
    return out


# front-end configurations (audio_model.py:39-43,149-151,257-260,367-369,515-516)
#
# [3P] tiny_L3 passes `n_win=480` to kapre's Spectrogram (audio_model.py:507-516: n_dft=512, n_win=480,
# n_hop=n_win//2=240).  Neither pinned kapre (0.1.3.1 in requirements_cpu.txt, 0.1.4 in requirements.txt) has an
# `n_win` argument in `Spectrogram.__init__(n_dft, n_hop, padding, power_spectrogram, return_decibel_spectrogram,
# trainable_kernel, image_data_format, **kwargs)`; the extra keyword travels in **kwargs to keras `Layer.__init__`,
# which in keras 2.0.9 rejects unknown keywords ("Keyword argument not understood") -- as far as can be told
# without the packages, `MODELS['tiny_L3']()` does not construct under the pinned versions, and no reference
# artefact (notebook shape, weight file) shows what window was meant.  Decision restated here and in
# engine.hip (FE_TINY): the DFT window is the kapre default, a periodic Hann of n_dft = 512 samples (n_win is
# ignored), hop 240, 'valid' framing -> (48000 - 512) // 240 + 1 = 198 frames.  A later kapre (>= 0.1.5) that
# accepts n_win would window 480 samples zero-padded to 512; that variant is NOT what is implemented.
FRONTENDS = {
    'orig':           dict(n_dft=512, n_hop=242, padding='valid', n_mels=0, power=1.0, db=False, loglambda=True),
    'kapredb':        dict(n_dft=512, n_hop=242, padding='valid', n_mels=0, power=1.0, db=True, loglambda=False),
    'melspec1':       dict(n_dft=2048, n_hop=242, padding='same', n_mels=128, power=1.0, db=True, loglambda=False),
    'melspec2':       dict(n_dft=2048, n_hop=242, padding='same', n_mels=256, power=1.0, db=True, loglambda=False),
    'tiny':           dict(n_dft=512, n_hop=240, padding='valid', n_mels=0, power=2.0, db=True, loglambda=False),
}


def tf_same_pad(n_in, k, s):
    """[3P] TensorFlow 'SAME' padding: out = ceil(n/s); total = max((out-1)s+k-n,0);
    before = total // 2."""
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return out, total // 2, total - total // 2


def frame_signal(audio, n_dft, n_hop, padding):
    """Frames of the strided conv2d kapre uses for the STFT.  audio (B,1,T)."""
    B, _, T = audio.shape
    x = audio[:, 0, :]
    if padding == 'same':
        n_frames, pl, pr = tf_same_pad(T, n_dft, n_hop)
        x = np.pad(x, ((0, 0), (pl, pr)))
    else:
        n_frames = (T - n_dft) // n_hop + 1
    idx = np.arange(n_frames)[:, None] * n_hop + np.arange(n_dft)[None, :]
    return x[:, idx]                       # (B, n_frames, n_dft)


def amplitude_to_decibel(x, scope='sample', amin=1e-10, dynamic_range=80.0):
    """[3P] kapre.backend_keras.amplitude_to_decibel.  0.1.4 subtracts the max
    per sample (scope='sample'); 0.1.3.1 is believed to subtract the batch-wide
    max (scope='batch')."""
    ln10 = np.asarray(np.log(10.0), dtype=x.dtype)
    log_spec = 10 * np.log(np.maximum(x, np.asarray(amin, dtype=x.dtype))) / ln10
    if scope == 'sample':
        mx = log_spec.reshape(log_spec.shape[0], -1).max(axis=1)
        mx = mx.reshape((-1,) + (1,) * (log_spec.ndim - 1))
    else:
        mx = log_spec.max()
    log_spec = log_spec - mx
    return np.maximum(log_spec, np.asarray(-dynamic_range, dtype=x.dtype))


def frontend_forward(kind, audio, consts=None, db_max_scope='sample', dtype=np.float64):
    """audio (B,1,48000) -> (B, n_freq|n_mels, n_frames, 1)."""
    cfg = FRONTENDS[kind]
    if consts is None:
        consts = frontend_constants(kind)
    real = consts['real_kernels'].reshape(cfg['n_dft'], -1).astype(dtype)
    imag = consts['imag_kernels'].reshape(cfg['n_dft'], -1).astype(dtype)
    fr = frame_signal(audio.astype(dtype), cfg['n_dft'], cfg['n_hop'], cfg['padding'])
    B, nf, _ = fr.shape
    fr2 = fr.reshape(B * nf, -1)
    re = fr2 @ real
    im = fr2 @ imag
    p = (re * re + im * im).reshape(B, nf, -1)              # (B, time, freq) power
    if cfg['n_mels']:
        p = p @ consts['freq2mel'].astype(dtype)            # (B, time, mel)
    if cfg['power'] != 2.0:
        p = np.power(np.sqrt(p), np.asarray(cfg['power'], dtype=dtype))
    out = np.transpose(p, (0, 2, 1))[..., None]             # (B, freq, time, 1)
    if cfg['db']:
        out = amplitude_to_decibel(out, scope=db_max_scope)
    if cfg['loglambda']:
        # audio_model.py:43  tf.log(tf.maximum(x, 1e-12)) / 5.0
        out = np.log(np.maximum(out, np.asarray(1e-12, dtype=dtype))) / np.asarray(5.0, dtype=dtype)
    return np.ascontiguousarray(out)


# ----------------------------------------------------------------------------
# primitive ops (forward + backward), NHWC
# ----------------------------------------------------------------------------
def _conv_pads(H, W, kh, kw, padding):
    if padding == 'same':
        Ho, pt, pb = tf_same_pad(H, kh, 1)
        Wo, pl, pr = tf_same_pad(W, kw, 1)
    else:
        Ho, Wo, pt, pb, pl, pr = H - kh + 1, W - kw + 1, 0, 0, 0, 0
    return Ho, Wo, pt, pb, pl, pr


# ---- mixed precision (BASELINE.json configs[4]: "bf16 compute / fp32 accumulate") ------------------
# Not in the reference (fp32 throughout); restated here so the build's L3_DTYPE_BF16 mode has an
# oracle.  Rules (same as l3embedding_amd/csrc/conv_bf16.hip and engine.hip):
#  (1) a 3x3 'same' convolution whose Cin and Cout are multiples of 64 rounds BOTH operands of its
#      forward, data-gradient and weight-gradient products to bfloat16 (round to nearest even) and
#      accumulates in the working dtype.  bf16 x bf16 products are exact in float32, so only the
#      summation order differs from the kernels;
#  (2) inside a tower, such a convolution -- and the tower's first convolution (3x3 'same', 1 or 3 input
#      channels, 64 filters; fp32 arithmetic) -- that feeds a BatchNormalization (directly or through the
#      Activation of vision_model.py:137-139) STORES its output (accumulator + bias) as bfloat16, and
#      the BatchNorm -- statistics, normalisation and both gradients -- is the ordinary fp32 math on that
#      stored tensor (the rounding is a straight-through identity for the gradient).  Exception: the
#      '<tower>_embedding_layer' output stays unrounded, because load_embedding() max-pools it directly
#      (audio_model.py:482-483, vision_model.py:212-215).
#  (3) the data gradient such a convolution writes -- the gradient at the BatchNorm (+ pool) output that feeds
#      it, read only by that BatchNorm's backward -- is stored as bfloat16 as well.
# Everything else is untouched.
CONV_OPERANDS = None          # None | 'bf16'


class mixed_precision(object):
    """`with mixed_precision('bf16'): ...` -- conv operand rounding for the enclosed oracle calls."""

    def __init__(self, kind):
        self.kind = None if kind in (None, 'f32', 'fp32', 'float32') else kind

    def __enter__(self):
        global CONV_OPERANDS
        self.prev = CONV_OPERANDS
        CONV_OPERANDS = self.kind

    def __exit__(self, *exc):
        global CONV_OPERANDS
        CONV_OPERANDS = self.prev


def bf16_round(a):
    """float -> nearest bfloat16 (ties to even), returned in a's dtype."""
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    u = a32.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(a.dtype)


def _stores_bf16(kh, kw, ci, co, padding):
    """Rule (2): which tower convolutions store their output as bfloat16 in mixed-precision mode -- the
    mixed-precision convolutions themselves and the first convolution of a tower (3x3 'same', 1 or 3 input
    channels, 64 filters: computed in fp32, its 64-channel full-resolution output is the largest activation)."""
    if CONV_OPERANDS != 'bf16':
        return False
    return _mp_conv(kh, kw, ci, co, padding) or (kh == 3 and kw == 3 and padding == 'same' and ci in (1, 3) and co == 64)


def _mp_conv(kh, kw, ci, co, padding):
    return CONV_OPERANDS == 'bf16' and kh == 3 and kw == 3 and padding == 'same' and ci % 64 == 0 and co % 64 == 0


def conv2d_fwd(x, w, b, padding):
    """stride-1 Conv2D + bias, tap-wise accumulation of (pixels,Cin)@(Cin,Cout)."""
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    if _mp_conv(kh, kw, Ci, Co, padding):
        x, w = bf16_round(x), bf16_round(w)
    Ho, Wo, pt, pb, pl, pr = _conv_pads(H, W, kh, kw, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((B * Ho * Wo, Co), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            xs = xp[:, i:i + Ho, j:j + Wo, :].reshape(-1, Ci)
            out += xs @ w[i, j]
    out += b
    return out.reshape(B, Ho, Wo, Co)


def conv2d_bwd(x, w, dy, padding, need_dx=True):
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    db = dy.reshape(-1, Co).sum(axis=0)          # bias gradient: a plain fp32 column sum, never rounded
    if _mp_conv(kh, kw, Ci, Co, padding):
        x, w, dy = bf16_round(x), bf16_round(w), bf16_round(dy)
    Ho, Wo, pt, pb, pl, pr = _conv_pads(H, W, kh, kw, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dy2 = dy.reshape(-1, Co)
    dw = np.zeros_like(w)
    dxp = np.zeros_like(xp) if need_dx else None
    for i in range(kh):
        for j in range(kw):
            xs = xp[:, i:i + Ho, j:j + Wo, :].reshape(-1, Ci)
            dw[i, j] = xs.T @ dy2
            if need_dx:
                dxp[:, i:i + Ho, j:j + Wo, :] += (dy2 @ w[i, j].T).reshape(B, Ho, Wo, Ci)
    dx = dxp[:, pt:pt + H, pl:pl + W, :] if need_dx else None
    return dx, dw, db


def bn_fwd(x, gamma, beta, mov_mean, mov_var, training):
    """[3P] keras BatchNormalization(axis=-1): batch moments (biased variance)
    in training, moving statistics otherwise."""
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    if training:
        mean = x2.mean(axis=0)
        var = ((x2 - mean) ** 2).mean(axis=0)
    else:
        mean, var = mov_mean.astype(x.dtype), mov_var.astype(x.dtype)
    rstd = 1.0 / np.sqrt(var + np.asarray(BN_EPS, dtype=x.dtype))
    xhat = (x2 - mean) * rstd
    y = (xhat * gamma + beta).reshape(x.shape)
    return y, (xhat, rstd, mean, var)


def bn_bwd(dy, gamma, cache, training):
    xhat, rstd, _, _ = cache
    C = dy.shape[-1]
    dy2 = dy.reshape(-1, C)
    dgamma = (dy2 * xhat).sum(axis=0)
    dbeta = dy2.sum(axis=0)
    if training:
        n = dy2.shape[0]
        dx = (gamma * rstd) * (dy2 - dbeta / n - xhat * (dgamma / n))
    else:
        dx = dy2 * (gamma * rstd)
    return dx.reshape(dy.shape), dgamma, dbeta


def _pool_geometry(H, W, ph, pw, sh, sw, padding):
    if padding == 'same':
        Ho, pt, _ = tf_same_pad(H, ph, sh)
        Wo, pl, _ = tf_same_pad(W, pw, sw)
    else:
        Ho, Wo, pt, pl = (H - ph) // sh + 1, (W - pw) // sw + 1, 0, 0
    return Ho, Wo, pt, pl


def maxpool_fwd(x, ph, pw, sh, sw, padding):
    """[3P] MaxPooling2D; 'same' pads with -inf (TF semantics).  Returns the
    flat (row-major scan, first max wins) argmax inside each window."""
    B, H, W, C = x.shape
    Ho, Wo, pt, pl = _pool_geometry(H, W, ph, pw, sh, sw, padding)
    Hp = max((Ho - 1) * sh + ph, H + pt)
    Wp = max((Wo - 1) * sw + pw, W + pl)
    xp = np.full((B, Hp, Wp, C), -np.inf, dtype=x.dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    best = np.full((B, Ho, Wo, C), -np.inf, dtype=x.dtype)
    arg = np.zeros((B, Ho, Wo, C), dtype=np.int32)
    for i in range(ph):
        for j in range(pw):
            v = xp[:, i:i + (Ho - 1) * sh + 1:sh, j:j + (Wo - 1) * sw + 1:sw, :]
            m = v > best
            best = np.where(m, v, best)
            arg = np.where(m, i * pw + j, arg)
    return best, (arg, (B, H, W, C), (ph, pw, sh, sw, pt, pl))


def maxpool_bwd(dy, cache):
    arg, (B, H, W, C), (ph, pw, sh, sw, pt, pl) = cache
    _, Ho, Wo, _ = dy.shape
    Hp = max((Ho - 1) * sh + ph, H + pt)
    Wp = max((Wo - 1) * sw + pw, W + pl)
    dxp = np.zeros((B, Hp, Wp, C), dtype=dy.dtype)
    for i in range(ph):
        for j in range(pw):
            m = (arg == i * pw + j)
            dxp[:, i:i + (Ho - 1) * sh + 1:sh, j:j + (Wo - 1) * sw + 1:sw, :] += np.where(m, dy, 0)
    return dxp[:, pt:pt + H, pl:pl + W, :]


# ----------------------------------------------------------------------------
# model ledger (A1-A4, A8): op lists per tower, keras auto-names
# ----------------------------------------------------------------------------
def _vgg_blocks(ops, counters, emb_name, last_pool, pool_padding, quirk_relu_bn):
    """4 x [Conv3x3 BN ReLU]x2 + MaxPool (audio_model.py:372-435, vision_model.py:126-189)."""
    filters = [64, 128, 256, 512]
    for bi, f in enumerate(filters):
        for ci in range(2):
            if bi == 3 and ci == 1:
                cname = emb_name
            else:
                counters['conv2d'] += 1
                cname = 'conv2d_%d' % counters['conv2d']
            counters['batch_normalization'] += 1
            bname = 'batch_normalization_%d' % counters['batch_normalization']
            ops.append(('conv', cname, f, 3, 3, 'same'))
            if quirk_relu_bn and bi == 0 and ci == 1:
                # vision_model.py:138-139 / 42-43: Activation THEN BatchNormalization
                ops.append(('relu',))
                ops.append(('bn', bname))
            else:
                ops.append(('bn', bname))
                ops.append(('relu',))
        if bi < 3:
            ops.append(('pool', 2, 2, 2, 2, pool_padding))
        else:
            ops.append(('pool',) + last_pool + last_pool + (pool_padding,))
    ops.append(('flatten',))


def _tiny_blocks(ops, counters):
    for _ in range(3):
        counters['conv2d'] += 1
        counters['batch_normalization'] += 1
        ops.append(('conv', 'conv2d_%d' % counters['conv2d'], 10, 5, 5, 'valid'))
        ops.append(('bn', 'batch_normalization_%d' % counters['batch_normalization']))
        ops.append(('relu',))
        ops.append(('pool', 3, 3, 3, 3, 'valid'))
    ops.append(('flatten',))


MODEL_TYPES = ('cnn_L3_orig', 'tiny_L3', 'cnn_L3_kapredbinputbn', 'cnn_L3_melspec1', 'cnn_L3_melspec2')


def model_spec(model_type):
    """Layer ledger for a registry entry of model.py:307-313.  Vision tower is
    constructed first (model.py:214-215,...,280-281) so it takes the lower keras
    auto-name indices."""
    if model_type not in MODEL_TYPES:
        raise ValueError('Invalid model type: "{}"'.format(model_type))
    counters = {'conv2d': 0, 'batch_normalization': 0}
    vis, aud = [], []
    if model_type == 'tiny_L3':
        _tiny_blocks(vis, counters)
        frontend, fe_name = 'tiny', 'spectrogram_1'
        _tiny_blocks(aud, counters)
        head = 64
    else:
        inputbn = model_type != 'cnn_L3_orig'
        if inputbn:
            counters['batch_normalization'] += 1
            vis.append(('bn', 'batch_normalization_%d' % counters['batch_normalization']))
        _vgg_blocks(vis, counters, 'vision_embedding_layer', (28, 28), 'same', True)
        frontend = {'cnn_L3_orig': 'orig', 'cnn_L3_kapredbinputbn': 'kapredb',
                    'cnn_L3_melspec1': 'melspec1', 'cnn_L3_melspec2': 'melspec2'}[model_type]
        fe_name = 'melspectrogram_1' if frontend.startswith('mel') else 'spectrogram_1'
        if inputbn:
            counters['batch_normalization'] += 1
            aud.append(('bn', 'batch_normalization_%d' % counters['batch_normalization']))
        last_pool = (16, 24) if model_type == 'cnn_L3_melspec1' else (32, 24)
        _vgg_blocks(aud, counters, 'audio_embedding_layer', last_pool, 'valid', False)
        head = 128
    return dict(model_type=model_type, vision=vis, audio=aud, frontend=frontend,
                frontend_name=fe_name, head=head)


def frontend_out_shape(kind):
    cfg = FRONTENDS[kind]
    if cfg['padding'] == 'same':
        nf = tf_same_pad(48000, cfg['n_dft'], cfg['n_hop'])[0]
    else:
        nf = (48000 - cfg['n_dft']) // cfg['n_hop'] + 1
    nfreq = cfg['n_mels'] if cfg['n_mels'] else cfg['n_dft'] // 2 + 1
    return (nfreq, nf, 1)


def tower_shapes(ops, in_shape):
    """Output (H,W,C) after every op."""
    H, W, C = in_shape
    shapes = []
    for op in ops:
        if op[0] == 'conv':
            _, _, f, kh, kw, padding = op
            H, W = _conv_pads(H, W, kh, kw, padding)[:2]
            C = f
        elif op[0] == 'pool':
            _, ph, pw, sh, sw, padding = op
            H, W = _pool_geometry(H, W, ph, pw, sh, sw, padding)[:2]
        elif op[0] == 'flatten':
            H, W, C = 1, 1, H * W * C
        shapes.append((H, W, C))
    return shapes


def param_table(model_type):
    """[(name, shape, trainable, kind)] in keras get_weights() order: layer by
    layer, kapre tensors first in the audio tower.  kind in {kernel,bias,gamma,
    beta,moving_mean,moving_variance,const}."""
    spec = model_spec(model_type)
    tab = []

    def tower(prefix, ops, in_c):
        c = in_c
        for op in ops:
            if op[0] == 'conv':
                _, name, f, kh, kw, _ = op
                tab.append(('%s/%s/kernel' % (prefix, name), (kh, kw, c, f), True, 'kernel'))
                tab.append(('%s/%s/bias' % (prefix, name), (f,), True, 'bias'))
                c = f
            elif op[0] == 'bn':
                name = op[1]
                tab.append(('%s/%s/gamma' % (prefix, name), (c,), True, 'gamma'))
                tab.append(('%s/%s/beta' % (prefix, name), (c,), True, 'beta'))
                tab.append(('%s/%s/moving_mean' % (prefix, name), (c,), False, 'moving_mean'))
                tab.append(('%s/%s/moving_variance' % (prefix, name), (c,), False, 'moving_variance'))

    tower('vision_model', spec['vision'], 3)
    cfg = FRONTENDS[spec['frontend']]
    nb = cfg['n_dft'] // 2 + 1
    fe = 'audio_model/' + spec['frontend_name']
    tab.append((fe + '/real_kernels', (cfg['n_dft'], 1, 1, nb), False, 'const'))
    tab.append((fe + '/imag_kernels', (cfg['n_dft'], 1, 1, nb), False, 'const'))
    if cfg['n_mels']:
        tab.append((fe + '/freq2mel', (nb, cfg['n_mels']), False, 'const'))
    tower('audio_model', spec['audio'], 1)
    v_out = tower_shapes(spec['vision'], (224, 224, 3))[-1][2]
    a_out = tower_shapes(spec['audio'], frontend_out_shape(spec['frontend']))[-1][2]
    tab.append(('dense_1/kernel', (v_out + a_out, spec['head']), True, 'kernel'))
    tab.append(('dense_1/bias', (spec['head'],), True, 'bias'))
    tab.append(('dense_2/kernel', (spec['head'], 2), True, 'kernel'))
    tab.append(('dense_2/bias', (2,), True, 'bias'))
    return tab


def truncated_normal(rng, shape, stddev):
    """[3P] tf.truncated_normal: N(0, stddev) with draws beyond two standard deviations re-drawn."""
    z = rng.standard_normal(shape)
    bad = np.abs(z) > 2.0
    while bad.any():
        z[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(z) > 2.0
    return z * stddev


def init_params(model_type, seed=20180123, dtype=np.float32):
    """he_normal kernels ([3P] keras 2.0.9 `he_normal` = VarianceScaling(scale=2, mode='fan_in',
    distribution='normal'), which draws from K.truncated_normal(stddev = sqrt(2 / fan_in)): values beyond
    2 sigma are re-drawn, so the effective standard deviation is ~0.88 sigma), zero biases, BN gamma=1 beta=0
    mean=0 var=1.  The engine initialises the same way (engine.hip alloc_everything) from its own generator."""
    rng = np.random.RandomState(seed)
    spec = model_spec(model_type)
    consts = frontend_constants(spec['frontend'])
    P = OrderedDict()
    for name, shape, _, kind in param_table(model_type):
        if kind == 'kernel':
            fan_in = int(np.prod(shape[:-1]))
            P[name] = truncated_normal(rng, shape, np.sqrt(2.0 / fan_in)).astype(dtype)
        elif kind in ('bias', 'beta', 'moving_mean'):
            P[name] = np.zeros(shape, dtype=dtype)
        elif kind in ('gamma', 'moving_variance'):
            P[name] = np.ones(shape, dtype=dtype)
        else:
            P[name] = consts[name.rsplit('/', 1)[1]].reshape(shape).astype(dtype)
    return P


# ----------------------------------------------------------------------------
# forward / backward of the whole AVC graph
# ----------------------------------------------------------------------------
def _feeds_batchnorm(ops, k):
    nxt = [o[0] for o in ops[k + 1:k + 3]]
    return nxt[:1] == ['bn'] or nxt == ['relu', 'bn']


def _tower_forward(prefix, ops, x, P, training, taps=None):
    caches = []
    for k, op in enumerate(ops):
        if op[0] == 'conv':
            name, padding = op[1], op[5]
            w = P['%s/%s/kernel' % (prefix, name)].astype(x.dtype)
            b = P['%s/%s/bias' % (prefix, name)].astype(x.dtype)
            y = conv2d_fwd(x, w, b, padding)
            if _stores_bf16(op[3], op[4], x.shape[-1], op[2], padding) and _feeds_batchnorm(ops, k) and \
                    not name.endswith('_embedding_layer'):
                y = bf16_round(y)            # mixed-precision rule (2): the conv output is stored as bfloat16
            caches.append((x, w))
            if taps is not None:
                taps[name] = y
            x = y
        elif op[0] == 'bn':
            name = op[1]
            g = P['%s/%s/gamma' % (prefix, name)].astype(x.dtype)
            bt = P['%s/%s/beta' % (prefix, name)].astype(x.dtype)
            mm = P['%s/%s/moving_mean' % (prefix, name)]
            mv = P['%s/%s/moving_variance' % (prefix, name)]
            x, c = bn_fwd(x, g, bt, mm, mv, training)
            caches.append((g, c))
        elif op[0] == 'relu':
            caches.append(x > 0)
            x = np.maximum(x, 0)
        elif op[0] == 'pool':
            x, c = maxpool_fwd(x, *op[1:])
            caches.append(c)
        elif op[0] == 'flatten':
            caches.append(x.shape)
            x = x.reshape(x.shape[0], -1)
    return x, caches


def _tower_backward(prefix, ops, dy, caches, training, G, need_input_grad=False):
    n = len(ops)
    for k in range(n - 1, -1, -1):
        op, c = ops[k], caches[k]
        if op[0] == 'flatten':
            dy = dy.reshape(c)
        elif op[0] == 'pool':
            dy = maxpool_bwd(dy, c)
        elif op[0] == 'relu':
            dy = np.where(c, dy, 0)
        elif op[0] == 'bn':
            g, bc = c
            dy, dg, db = bn_bwd(dy, g, bc, training)
            G['%s/%s/gamma' % (prefix, op[1])] = dg
            G['%s/%s/beta' % (prefix, op[1])] = db
        elif op[0] == 'conv':
            x, w = c
            need_dx = need_input_grad or any(o[0] in ('conv', 'bn') for o in ops[:k])
            dy, dw, db = conv2d_bwd(x, w, dy, op[5], need_dx=need_dx)
            if need_dx and _mp_conv(op[3], op[4], x.shape[-1], op[2], op[5]):
                dy = bf16_round(dy)          # mixed-precision rule (3): the data gradient is stored as bfloat16
            G['%s/%s/kernel' % (prefix, op[1])] = dw
            G['%s/%s/bias' % (prefix, op[1])] = db
    return dy


def forward(model_type, P, video, audio, training, dtype=np.float64,
            db_max_scope='sample', want_taps=False):
    """Returns dict(probs, logits, v, a, spec_in, caches...)."""
    spec = model_spec(model_type)
    consts = {k.rsplit('/', 1)[1]: v for k, v in P.items() if '/' + spec['frontend_name'] + '/' in k}
    taps = {} if want_taps else None
    xv = video.astype(dtype)
    fe = frontend_forward(spec['frontend'], audio, consts, db_max_scope, dtype)
    v, cv = _tower_forward('vision_model', spec['vision'], xv, P, training, taps)
    a, ca = _tower_forward('audio_model', spec['audio'], fe, P, training, taps)
    h0 = np.concatenate([v, a], axis=1)          # model.py:25 vision first
    w1, b1 = P['dense_1/kernel'].astype(dtype), P['dense_1/bias'].astype(dtype)
    w2, b2 = P['dense_2/kernel'].astype(dtype), P['dense_2/bias'].astype(dtype)
    z1 = h0 @ w1 + b1
    h1 = np.maximum(z1, 0)
    logits = h1 @ w2 + b2
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    probs = e / e.sum(axis=1, keepdims=True)
    return dict(probs=probs, logits=logits, v=v, a=a, frontend=fe, h0=h0, z1=z1, h1=h1,
                cv=cv, ca=ca, w1=w1, w2=w2, taps=taps, spec=spec)


def l2_penalty(P, model_type, dtype=np.float64):
    """sum over the 18 (16 conv + 2 dense) kernels of 1e-5 * sum(w^2)  [3P keras
    regularizers.l2: l2 * sum(square(w)), no 1/2]."""
    tot = dtype(0)
    for name, _, _, kind in param_table(model_type):
        if kind == 'kernel':
            w = P[name].astype(dtype)
            tot += dtype(L2_WEIGHT) * (w * w).sum()
    return tot


def loss_and_grads(model_type, P, video, audio, labels, training=True, dtype=np.float64,
                   db_max_scope='sample'):
    """[3P] keras categorical_crossentropy on softmax probabilities:
    p /= sum(p); p = clip(p, 1e-7, 1-1e-7); loss = mean_B(-sum_c t_c log p_c)
    + L2 penalties; metric acc = mean(argmax p == argmax t).  Returns
    (out dict, grads OrderedDict over trainable params)."""
    f = forward(model_type, P, video, audio, training, dtype, db_max_scope)
    t = labels.astype(dtype)
    p = f['probs']
    B = p.shape[0]
    s = p.sum(axis=1, keepdims=True)
    q = p / s
    eps = dtype(K_EPSILON)
    qc = np.clip(q, eps, 1 - eps)
    data_loss = (-(t * np.log(qc)).sum(axis=1)).mean()
    reg = l2_penalty(P, model_type, dtype)
    acc = (p.argmax(axis=1) == t.argmax(axis=1)).mean()
    # backward: loss -> qc -> q -> p -> logits
    dqc = -(t / qc) / B
    dq = np.where((q >= eps) & (q <= 1 - eps), dqc, 0)
    dp = dq / s - (dq * p).sum(axis=1, keepdims=True) / (s * s)
    dlogits = p * (dp - (dp * p).sum(axis=1, keepdims=True))
    G = OrderedDict()
    G['dense_2/kernel'] = f['h1'].T @ dlogits
    G['dense_2/bias'] = dlogits.sum(axis=0)
    dh1 = dlogits @ f['w2'].T
    dz1 = np.where(f['z1'] > 0, dh1, 0)
    G['dense_1/kernel'] = f['h0'].T @ dz1
    G['dense_1/bias'] = dz1.sum(axis=0)
    dh0 = dz1 @ f['w1'].T
    nv = f['v'].shape[1]
    spec = f['spec']
    _tower_backward('vision_model', spec['vision'], dh0[:, :nv], f['cv'], training, G)
    _tower_backward('audio_model', spec['audio'], dh0[:, nv:], f['ca'], training, G)
    for name, _, _, kind in param_table(model_type):
        if kind == 'kernel':
            G[name] = G[name] + 2 * dtype(L2_WEIGHT) * P[name].astype(dtype)
    out = dict(loss=data_loss + reg, data_loss=data_loss, reg=reg, acc=acc, probs=p,
               logits=f['logits'], fwd=f)
    grads = OrderedDict((n, G[n]) for n, _, tr, _ in param_table(model_type) if tr)
    return out, grads


def bn_batch_stats(model_type, fwd):
    """{bn layer full name: (batch mean, batch var)} of a training-mode forward."""
    stats = {}
    spec = fwd['spec']
    for prefix, ops, caches in (('vision_model', spec['vision'], fwd['cv']),
                                ('audio_model', spec['audio'], fwd['ca'])):
        for op, c in zip(ops, caches):
            if op[0] == 'bn':
                stats['%s/%s' % (prefix, op[1])] = (c[1][2], c[1][3])
    return stats


class AdamState(object):
    """[3P] keras.optimizers.Adam (lr, 0.9, 0.999, 1e-8, decay 0):
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps), t from 1."""

    def __init__(self):
        self.t = 0
        self.m = {}
        self.v = {}


def adam_update(P, grads, state, lr, dtype=np.float64):
    state.t += 1
    t = state.t
    lr_t = dtype(lr) * np.sqrt(1.0 - dtype(ADAM_B2) ** t) / (1.0 - dtype(ADAM_B1) ** t)
    for name, g in grads.items():
        g = g.astype(dtype)
        m = state.m.get(name, np.zeros_like(g))
        v = state.v.get(name, np.zeros_like(g))
        m = dtype(ADAM_B1) * m + (1 - dtype(ADAM_B1)) * g
        v = dtype(ADAM_B2) * v + (1 - dtype(ADAM_B2)) * g * g
        state.m[name], state.v[name] = m, v
        P[name] = (P[name].astype(dtype) - lr_t * m / (np.sqrt(v) + dtype(ADAM_EPS))).astype(P[name].dtype)


class BNMovingState(object):
    """[3P] keras 2.0.9 TF backend moving_average_update = tf
    assign_moving_average(x, value, momentum, zero_debias=True): hidden `biased`
    (init 0) and `local_step` accumulators; variable <- biased / (1 - m^step).
    zero_debias=False gives the plain EMA  x <- x*m + value*(1-m)."""

    def __init__(self, zero_debias=True):
        self.zero_debias = zero_debias
        self.biased = {}
        self.step = {}

    def update(self, P, name, value):
        mom = BN_MOMENTUM
        if self.zero_debias:
            b = self.biased.get(name, np.zeros_like(value, dtype=np.float64))
            s = self.step.get(name, 0) + 1
            b = b - (b - value.astype(np.float64)) * (1 - mom)
            self.biased[name], self.step[name] = b, s
            P[name] = (b / (1 - mom ** s)).astype(P[name].dtype)
        else:
            P[name] = (P[name].astype(np.float64) * mom + value.astype(np.float64) * (1 - mom)).astype(P[name].dtype)


def dp_train_step(model_type, P, adam, bnstate, video, audio, labels, lr, world, dtype=np.float64, db_max_scope='sample',
                  moving='replicas'):
    """One step of multi_gpu_model(model, gpus=world) (training_utils.py:121-170) on `world` virtual replicas.

    The global batch is cut with get_slice's arithmetic (:121-133); the template model is CALLED once per replica (:155
    `outputs = model(inputs)`): every replica normalises with the batch statistics of ITS slice (no sync-BN) and each
    BatchNormalization call adds its own moving-average update of the one shared variable -- `world` updates per layer and step,
    applied here in replica order 0..world-1 (the order the loop :141 builds them; [3P] TF does not order them among themselves).
    The outputs are concatenated (:165-170) and the loss is the mean over the global batch, so the gradient is
    sum_r (n_r / B) * grad_r (+ the L2 penalty once).  moving='rank_local': only replica 0's update (what a rank that keeps its
    own statistics stores).  Mutates P, adam, bnstate; returns the per-replica forward outputs."""
    B = len(labels)
    names = [n for n, _, t, _ in param_table(model_type) if t]
    total = {n: 0.0 for n in names}
    outs, stats_all, loss = [], [], 0.0
    for r in range(world):
        lo, hi = dp_slice(B, world, r)
        out, g = loss_and_grads(model_type, P, video[lo:hi], audio[lo:hi], labels[lo:hi], True, dtype, db_max_scope)
        w = (hi - lo) / float(B)
        for n in names:
            reg = 2 * L2_WEIGHT * P[n].astype(np.float64) if n.endswith('/kernel') else 0.0
            total[n] = total[n] + (g[n] - reg) * w
        loss += out['data_loss'] * w
        stats_all.append(bn_batch_stats(model_type, out['fwd']))
        outs.append(out)
    for n in names:
        if n.endswith('/kernel'):
            total[n] = total[n] + 2 * L2_WEIGHT * P[n].astype(np.float64)
    adam_update(P, total, adam, lr, dtype)
    for r in range(world if moving == 'replicas' else 1):
        for lname, (mean, var) in stats_all[r].items():
            bnstate.update(P, lname + '/moving_mean', mean)
            bnstate.update(P, lname + '/moving_variance', var)
    return {'replicas': outs, 'grads': total, 'stats': stats_all, 'loss': loss + outs[0]['reg']}


def train_step(model_type, P, adam, bnstate, video, audio, labels, lr, dtype=np.float64,
               db_max_scope='sample'):
    """One fit_generator step (train.py:408-414): forward(training) -> loss ->
    backward -> Adam -> BN moving-stat update.  Mutates P, adam, bnstate."""
    out, grads = loss_and_grads(model_type, P, video, audio, labels, True, dtype, db_max_scope)
    stats = bn_batch_stats(model_type, out['fwd'])
    adam_update(P, grads, adam, lr, dtype)
    for lname, (mean, var) in stats.items():
        bnstate.update(P, lname + '/moving_mean', mean)
        bnstate.update(P, lname + '/moving_variance', var)
    out['grads'] = grads
    return out


# ----------------------------------------------------------------------------
# A7: data-parallel slicing (training_utils.py:121-133)
# ----------------------------------------------------------------------------
def dp_slice(batch_size, parts, i):
    """[start, stop) of replica i: step = B // parts, last replica takes the rest."""
    step = batch_size // parts
    start = step * i
    size = batch_size - step * i if i == parts - 1 else step
    return start, start + size


# ----------------------------------------------------------------------------
# A9: embedding extraction (model.py:131-181, audio_model.py:445-487,
#     vision_model.py:198-218)
# ----------------------------------------------------------------------------
AUDIO_POOLING = {
    'cnn_L3_orig': {'original': (8, 8), 'short': (32, 24)},
    'cnn_L3_kapredbinputbn': {'original': (8, 8), 'short': (32, 24)},
    'cnn_L3_melspec1': {'original': (4, 8), 'short': (16, 24)},
    'cnn_L3_melspec2': {'original': (8, 8), 'short': (32, 24)},
}


def embed_audio(model_type, P, audio, pooling_type='original', dtype=np.float64, db_max_scope='sample'):
    """MaxPooling2D(pool, padding='same') on the *conv output* of
    audio_embedding_layer (before its BN/ReLU), inference-mode BN, then Flatten."""
    spec = model_spec(model_type)
    ph, pw = AUDIO_POOLING[model_type][pooling_type]
    consts = {k.rsplit('/', 1)[1]: v for k, v in P.items() if '/' + spec['frontend_name'] + '/' in k}
    fe = frontend_forward(spec['frontend'], audio, consts, db_max_scope, dtype)
    taps = {}
    _tower_forward('audio_model', spec['audio'], fe, P, False, taps)
    y, _ = maxpool_fwd(taps['audio_embedding_layer'], ph, pw, ph, pw, 'same')
    return y.reshape(y.shape[0], -1)


def embed_vision(model_type, P, video, dtype=np.float64):
    spec = model_spec(model_type)
    taps = {}
    _tower_forward('vision_model', spec['vision'], video.astype(dtype), P, False, taps)
    y, _ = maxpool_fwd(taps['vision_embedding_layer'], 7, 7, 7, 7, 'same')
    return y.reshape(y.shape[0], -1)


def synthetic_batch(batch, seed=20180123, rank=0):
    """SURVEY 8(d) synthetic inputs: int16 PCM U{-32768..32767} through pcm2float,
    uint8 frames U{0..255} through preprocess_video, Bernoulli(0.5) one-hot labels."""
    rng = np.random.RandomState(seed + rank)
    pcm = rng.randint(-32768, 32768, size=(batch, 1, 48000)).astype(np.int16)
    frm = rng.randint(0, 256, size=(batch, 224, 224, 3)).astype(np.uint8)
    lab = rng.randint(0, 2, size=(batch,))
    labels = np.stack([lab, 1 - lab], axis=1).astype(np.float32)
    return preprocess_video(frm), pcm2float(pcm, np.float32), labels
