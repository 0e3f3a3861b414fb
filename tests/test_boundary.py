"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol
include/l3hip.h declares, the host mirror exposes the reference's names, and the product
path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from l3embedding_amd import _build, _lib, model
from l3embedding_amd.training_utils import get_slice_bounds, multi_gpu_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    _build.build()
    return _lib.load()


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'l3hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(l3_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol(lib):
    syms = header_symbols()
    assert len(syms) >= 35
    raw = ctypes.CDLL(_lib.lib_path())
    for s in syms:
        assert hasattr(raw, s), 'libl3hip.so does not export ' + s
    assert sorted(_lib.SIGNATURES) == syms, 'ctypes binding table and header disagree'


def test_model_registry_names_match_reference(lib):
    # model.py:307-313
    assert sorted(model.MODELS) == sorted(['cnn_L3_orig', 'tiny_L3', 'cnn_L3_kapredbinputbn', 'cnn_L3_melspec1', 'cnn_L3_melspec2'])
    for name, idx in _lib.MODEL_IDS.items():
        assert lib.l3_model_type_from_name(name.encode()) == idx
    assert lib.l3_model_type_from_name(b'cnn_L3_bogus') < 0
    with pytest.raises(ValueError, match='Invalid model type'):
        model.load_model('nowhere.h5', 'cnn_L3_bogus')                     # model.py:113-114
    m, inputs, out = model.MODELS['cnn_L3_melspec2']()
    assert m.name == 'cnn_L3_melspec2' and len(inputs) == 2
    assert inputs[0].shape == (None, 224, 224, 3) and inputs[1].shape == (None, 1, 48000)
    assert m.count_params() == 13977234
    assert m.get_layer('audio_model').count_params() == 9152708            # SURVEY 8(a) totals
    assert m.get_layer('vision_model').count_params() == 4693068
    m.get_layer('audio_model').get_layer('audio_embedding_layer')
    with pytest.raises(ValueError):
        m.get_layer('audio_model').get_layer('nope')
    with pytest.raises(ValueError, match='call `multi_gpu_model` with `gpus >= 2`'):
        multi_gpu_model(m, 1)                                              # training_utils.py:100-103


def test_multi_gpu_model_checks_the_devices_like_the_reference(lib):
    """training_utils.py:105-119: asking for more GPUs than the machine has is a ValueError naming both lists."""
    from l3embedding_amd.training_utils import available_devices
    have = available_devices()
    assert have[0] == '/cpu:0' and len(have) == 1 + lib.l3_device_count()
    m, _, _ = model.MODELS['tiny_L3']()
    want = len(have) - 1 + 2
    with pytest.raises(ValueError, match='we expect the following devices to be available') as ei:
        multi_gpu_model(m, want)
    assert '/gpu:%d' % (want - 1) in str(ei.value) and 'Try reducing `gpus`' in str(ei.value)
    assert m.replicas == 1
    # reading a wrapper-layout weight FILE needs no devices (model.load_model passes validate=False)
    assert multi_gpu_model(m, 8, validate=False).replicas == 8


def test_get_weights_follows_keras_order():
    """[3P] keras Model.get_weights(): top-level layers in order; a nested model contributes all its
    trainable tensors, then all its non-trainable ones (kapre constants + BatchNorm moving statistics) --
    the order save_weights writes.  A call on the sub-model itself is layer-interleaved with kapre's
    three constants first (notebooks/extract_spectrogram_models_from_avc_models.ipynb:446)."""
    from collections import OrderedDict
    m, _, _ = model.MODELS['cnn_L3_melspec2']()
    tab = m.param_table()
    m._host_weights = OrderedDict((n, np.full(s, i, np.float32)) for i, (n, s, _) in enumerate(tab))
    names = [n for n, _, _ in tab]
    got = [names[int(w.ravel()[0])] for w in m.get_weights()]
    assert len(got) == 111 and sorted(got) == sorted(names)
    tr = {n: t for n, _, t in tab}
    vis = [n for n in got if n.startswith('vision_model/')]
    aud = [n for n in got if n.startswith('audio_model/')]
    assert got == vis + aud + ['dense_1/kernel', 'dense_1/bias', 'dense_2/kernel', 'dense_2/bias']
    for sub in (vis, aud):
        flags = [tr[n] for n in sub]
        assert flags == sorted(flags, reverse=True)              # every trainable tensor before any non-trainable
    n_tr = sum(tr[n] for n in aud)
    assert aud[n_tr:n_tr + 3] == ['audio_model/melspectrogram_1/real_kernels', 'audio_model/melspectrogram_1/imag_kernels',
                                  'audio_model/melspectrogram_1/freq2mel']
    assert vis[0] == 'vision_model/batch_normalization_1/gamma' and vis[2] == 'vision_model/conv2d_1/kernel'
    sub_names = [names[int(w.ravel()[0])] for w in m.get_layer('audio_model').get_weights()]
    assert sub_names[:3] == aud[n_tr:n_tr + 3] and sub_names[3].endswith('batch_normalization_10/gamma')
    # set_weights(get_weights()) round-trips, and a wrong-length list is refused like keras does
    ws = m.get_weights()
    m.set_weights(ws)
    assert all(np.array_equal(a, b) for a, b in zip(ws, m.get_weights()))
    with pytest.raises(ValueError, match='expecting 111 weights'):
        m.set_weights(ws[:-1])


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.L3Error, match='not available'):
        _lib.Engine('cnn_L3_melspec2', 2)
    m, _, _ = model.MODELS['tiny_L3']()
    m.compile(model.Adam(lr=1e-4), loss='categorical_crossentropy', metrics=['accuracy'])
    with pytest.raises(_lib.L3Error):
        m.train_on_batch([np.zeros((1, 224, 224, 3), np.float32), np.zeros((1, 1, 48000), np.float32)],
                         np.array([[1, 0]], np.float32))
    with pytest.raises(_lib.L3Error):
        _lib.op_conv2d_fwd(np.zeros((1, 4, 4, 16), np.float32), np.zeros((3, 3, 16, 32), np.float32), np.zeros(32, np.float32), True)


def test_config_struct_layout():
    # must mirror `struct l3_config` in include/l3hip.h
    assert ctypes.sizeof(_lib.L3Config) == 48
    assert _lib.L3Config.stream.offset == 32 and _lib.L3Config.fp32_conv.offset == 40


def test_bad_create_arguments(lib):
    cfg = _lib.L3Config()
    cfg.struct_size = 4
    h = ctypes.c_void_p()
    assert lib.l3_create(ctypes.byref(cfg), 1, ctypes.byref(h)) == -1
    assert b'struct_size' in lib.l3_last_error(None)
    cfg.struct_size = ctypes.sizeof(_lib.L3Config)
    cfg.batch = 0
    assert lib.l3_create(ctypes.byref(cfg), 1, ctypes.byref(h)) == -1
    cfg.batch = 1
    cfg.model_type = 17
    assert lib.l3_create(ctypes.byref(cfg), 1, ctypes.byref(h)) == -1
    assert b'Invalid model type' in lib.l3_last_error(None)


def test_get_slice_bounds():
    assert [get_slice_bounds(512, 8, i) for i in (0, 7)] == [(0, 64), (448, 512)]
    assert [get_slice_bounds(10, 4, i) for i in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 10)]


def test_adam_defaults_only():
    assert model.Adam(lr=1e-5).lr == 1e-5
    with pytest.raises(ValueError):
        model.Adam(beta_1=0.5)


def test_tower_level_constructors_are_exported():
    """model.py:1-4 re-exports every public name of vision_model.py / audio_model.py: the tower constructors return
    (model, input, output) like the AVC-level ones, and L3_merge_audio_vision_models (model.py:7-35) joins a pair."""
    names = ['construct_cnn_L3_orig_vision_model', 'construct_cnn_L3_orig_inputbn_vision_model',          # vision_model.py:7,102
             'construct_cnn_l3_orig_vision_embedding_model', 'construct_tiny_L3_vision_model',            # :198,221
             'construct_cnn_L3_orig_audio_model', 'construct_cnn_L3_kapredbinputbn_audio_model',          # audio_model.py:8,118
             'construct_cnn_L3_melspec1_audio_model', 'construct_cnn_L3_melspec2_audio_model',            # :225,335
             'convert_audio_model_to_embedding', 'construct_tiny_L3_audio_model',                         # :445,490
             'L3_merge_audio_vision_models', 'convert_num_gpus', 'load_model', 'load_embedding', 'gpu_wrapper', 'MODELS']
    for n in names:
        assert callable(getattr(model, n)) or n == 'MODELS', n
    am, x_a, y_a = model.construct_cnn_L3_melspec2_audio_model()
    vm, x_i, y_i = model.construct_cnn_L3_orig_inputbn_vision_model()
    assert am.name == 'audio_model' and vm.name == 'vision_model'                 # audio_model.py:441, vision_model.py:194
    assert x_a.shape == (None, 1, 48000) and x_i.shape == (None, 224, 224, 3)
    assert am.output_shape == (None, 512) and vm.output_shape == (None, 512)
    # parameter counts of the notebooks (SURVEY.md 8(a) totals): vision 4,693,068 / audio 9,152,708 with the input BNs
    assert vm.count_params() == 4693068 and am.count_params() == 9152708
    assert am.get_layer('audio_embedding_layer').name == 'audio_embedding_layer'
    with pytest.raises(ValueError):
        am.get_layer('vision_embedding_layer')
    tv, _, _ = model.construct_tiny_L3_vision_model()
    ta, _, _ = model.construct_tiny_L3_audio_model()
    assert tv.output_shape == (None, 360) and ta.output_shape == (None, 350)
    # every registry entry is the merge of its two towers; anything else is refused
    for mt, (vk, ak) in model.TOWERS.items():
        v3 = getattr(model, 'construct_%s_model' % vk)()
        a3 = getattr(model, 'construct_%s_model' % ak)()
        m, inputs, y = model.L3_merge_audio_vision_models(v3[0], v3[1], a3[0], a3[1], mt, layer_size=64 if mt == 'tiny_L3' else 128)
        assert m.model_type == mt and m.name == mt and [i.shape for i in inputs] == [(None, 224, 224, 3), (None, 1, 48000)]
    with pytest.raises(ValueError):
        model.L3_merge_audio_vision_models(tv, None, am, None, 'x')
    with pytest.raises(ValueError):
        model.L3_merge_audio_vision_models(vm, x_i, am, x_a, 'cnn_L3_melspec2', layer_size=64)
