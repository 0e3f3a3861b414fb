"""Keras-2.0.9 HDF5 weight files written/read without h5py (l3embedding_amd.h5lite), cross-
checked against real libhdf5 through the container's conda h5py when it is present."""
import os
import subprocess
import sys

import numpy as np
import pytest

from l3embedding_amd import h5lite, kerasfile, model

CONDA_PY = '/opt/conda/bin/python3.9'


def _have_h5py():
    if not os.path.exists(CONDA_PY):
        return False
    return subprocess.call([CONDA_PY, '-c', 'import h5py'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 0


def _random_weights(mt, seed=0):
    rng = np.random.RandomState(seed)
    return {n: rng.randn(*s).astype(np.float32) for n, s, _ in model._host_param_table(mt)}


@pytest.mark.parametrize('mt,wrapper', [('cnn_L3_melspec2', False), ('cnn_L3_melspec2', True), ('tiny_L3', False), ('cnn_L3_orig', False)])
def test_roundtrip_own_reader(tmp_path, mt, wrapper):
    tab = model._host_param_table(mt)
    W = _random_weights(mt)
    path = str(tmp_path / 'model_latest.h5')
    kerasfile.save_weights(path, W, tab, mt, wrapper)
    R = kerasfile.load_weights(path, tab, mt, wrapper)
    assert set(R) == set(W)
    assert all(np.array_equal(R[k], W[k]) for k in W)
    root = h5lite.read_file(path)
    names = [n.decode() for n in root.attrs['layer_names']]
    if wrapper:
        assert mt in names and len(root[mt].attrs['weight_names']) == len(tab)      # model.py:77,117-119
    else:
        assert names == ['input_1', 'input_2', 'vision_model', 'audio_model', 'concatenate_1', 'dense_1', 'dense_2']
        wn = [n.decode() for n in root['vision_model'].attrs['weight_names']]
        assert wn[0].endswith('/gamma:0') or wn[0].endswith('/kernel:0')
        assert wn[-1].endswith('moving_variance:0')                                # trainable first, then non-trainable
        assert root['dense_1/dense_1/kernel:0'].shape == tab[-4][1]
    # loading a wrapper file into a plain model must fail like keras does
    with pytest.raises(ValueError):
        kerasfile.load_weights(path, tab, mt, not wrapper)


def test_rejects_non_hdf5(tmp_path):
    p = tmp_path / 'x.h5'
    p.write_bytes(b'not hdf5 at all')
    with pytest.raises(h5lite.H5Error):
        h5lite.read_file(str(p))


@pytest.mark.skipif(not _have_h5py(), reason='no h5py interpreter in this container')
def test_libhdf5_reads_our_files_and_we_read_h5py_files(tmp_path):
    mt = 'cnn_L3_melspec2'
    tab = model._host_param_table(mt)
    W = _random_weights(mt, 3)
    ours = str(tmp_path / 'ours.h5')
    theirs = str(tmp_path / 'theirs.h5')
    kerasfile.save_weights(ours, W, tab, mt, False)
    np.savez(str(tmp_path / 'w.npz'), **{k.replace('/', '__'): v for k, v in W.items()})
    # (1) real libhdf5 opens our file exactly the way keras 2.0.9 load_weights walks it, and
    # (2) writes the same weights the way keras 2.0.9 save_weights does (h5py defaults)
    script = r'''
import sys, h5py, numpy as np
ours, theirs, npz = sys.argv[1:4]
W = {k.replace('__', '/'): v for k, v in np.load(npz).items()}
f = h5py.File(ours, 'r')
layer_names = [n.decode('utf8') for n in f.attrs['layer_names']]
assert f.attrs['keras_version'] == b'2.0.9' and f.attrs['backend'] == b'tensorflow'
count = 0
for name in layer_names:
    g = f[name]
    wn = [n.decode('utf8') for n in g.attrs['weight_names']]
    for w in wn:
        a = np.asarray(g[w])
        key = (name + '/' + w[:-2]) if name in ('vision_model', 'audio_model') else w[:-2]
        key = key.replace('/Variable', '/freq2mel')
        assert np.array_equal(a, W[key]), key
        count += 1
assert count == len(W), (count, len(W))
out = h5py.File(theirs, 'w')
out.attrs['layer_names'] = [n.encode('utf8') for n in layer_names]
out.attrs['backend'] = 'tensorflow'.encode('utf8')
out.attrs['keras_version'] = '2.0.9'.encode('utf8')
for name in layer_names:
    g = out.create_group(name)
    wn = [n for n in f[name].attrs['weight_names']]
    g.attrs['weight_names'] = wn
    for n in wn:
        val = np.asarray(f[name][n.decode('utf8')])
        d = g.create_dataset(n.decode('utf8'), val.shape, dtype=val.dtype)
        d[:] = val
out.close()
print('OK', count)
'''
    res = subprocess.run([CONDA_PY, '-c', script, ours, theirs, str(tmp_path / 'w.npz')], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    R = kerasfile.load_weights(theirs, tab, mt, False)
    assert all(np.array_equal(R[k], W[k]) for k in W)
