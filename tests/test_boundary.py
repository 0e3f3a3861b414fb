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
    with pytest.raises(ValueError):
        multi_gpu_model(m, 1)                                              # training_utils.py:100-103


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
    assert ctypes.sizeof(_lib.L3Config) == 40
    assert _lib.L3Config.stream.offset == 32


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
