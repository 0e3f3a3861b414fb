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


# ---- name-first placement (l3embedding/model.py:77,117-119 load_weights; SURVEY 8(f)-1 "match by dataset name") -------
def _write_groups(path, mt, W, order_fn, rename=lambda w: w, wrapper=False):
    """A keras-layout file written with h5lite whose per-group dataset ORDER and NAMES are chosen by the test."""
    tab = model._host_param_table(mt)
    root = h5lite.Group()
    groups = kerasfile.keras_groups(tab, mt, wrapper)
    root.attrs['layer_names'] = np.array([g.encode() for g in groups])
    root.attrs['backend'] = b'tensorflow'
    root.attrs['keras_version'] = b'2.0.9'
    for gname, pnames in groups.items():
        g = root.create_group(gname)
        pn = order_fn(list(pnames))
        wn = [rename(kerasfile._tf_name(p)) for p in pn]
        g.attrs['weight_names'] = np.array([w.encode() for w in wn]) if wn else np.zeros((0,), np.float64)
        for p, w in zip(pn, wn):
            g.create_dataset(w, W[p])
    h5lite.write_file(path, root)


def _layer_interleaved(pnames):
    """kernel, bias, gamma, beta, moving_mean, moving_variance layer by layer (the ledger order), not
    trainable-then-non-trainable: what a nested model's own get_weights() order would give."""
    tab_order = {n: i for i, (n, _, _) in enumerate(model._host_param_table('cnn_L3_melspec2'))}
    return sorted(pnames, key=lambda n: tab_order[n])


def test_layer_interleaved_file_lands_in_named_slots(tmp_path):
    mt = 'cnn_L3_melspec2'
    tab = model._host_param_table(mt)
    W = _random_weights(mt, 5)
    path = str(tmp_path / 'interleaved.h5')
    _write_groups(path, mt, W, _layer_interleaved)
    rep = {}
    R = kerasfile.load_weights(path, tab, mt, report=rep)
    assert all(np.array_equal(R[k], W[k]) for k in W)
    assert set(rep.values()) == {'name'}
    # the same file read positionally would have put a moving_mean where gamma belongs: every BatchNorm tensor has shape (C,)
    pos = kerasfile.keras_groups(tab, mt)['vision_model']
    assert pos != _layer_interleaved(pos)
    # reversed order, and the tower prefix in the dataset names
    path2 = str(tmp_path / 'reversed.h5')
    _write_groups(path2, mt, W, lambda p: p[::-1])
    R = kerasfile.load_weights(path2, tab, mt)
    assert all(np.array_equal(R[k], W[k]) for k in W)


def test_shifted_auto_numbers_match_by_class_and_rank(tmp_path):
    """keras numbers layers per session: a model built after another one has conv2d_17.., batch_normalization_19.."""
    import re
    mt = 'cnn_L3_melspec2'
    tab = model._host_param_table(mt)
    W = _random_weights(mt, 6)

    def shifted(w):
        return re.sub(r'^(conv2d|batch_normalization|melspectrogram|dense)_(\d+)/',
                      lambda m: '%s_%d/' % (m.group(1), int(m.group(2)) + 16), w)

    path = str(tmp_path / 'shifted.h5')
    _write_groups(path, mt, W, _layer_interleaved, rename=shifted)
    rep = {}
    R = kerasfile.load_weights(path, tab, mt, report=rep)
    assert all(np.array_equal(R[k], W[k]) for k in W)
    assert rep['vision_model'] == 'class-rank' and rep['audio_model'] == 'class-rank'


def test_positional_fallback_refuses_a_permuted_kind(tmp_path):
    mt = 'tiny_L3'
    tab = model._host_param_table(mt)
    W = _random_weights(mt, 7)
    # names without a layer part: nothing to match by name -> position, in the writer's order: accepted
    path = str(tmp_path / 'bare.h5')
    _write_groups(path, mt, W, lambda p: p, rename=lambda w: w.split('/')[-1])
    with pytest.raises(ValueError):           # bare 'kernel:0' repeats inside a group
        kerasfile.load_weights(path, tab, mt)
    path = str(tmp_path / 'opaque.h5')
    cnt = [0]

    def opaque(w):
        cnt[0] += 1
        return 'w%d/%s' % (cnt[0], w.split('/')[-1])          # no class_number form, variable kind kept

    _write_groups(path, mt, W, lambda p: p, rename=opaque)
    rep = {}
    R = kerasfile.load_weights(path, tab, mt, report=rep)
    assert all(np.array_equal(R[k], W[k]) for k in W) and set(rep.values()) == {'position'}
    # ... but the same opaque names in layer-interleaved order would put moving statistics into gamma slots
    path = str(tmp_path / 'opaque_permuted.h5')
    _write_groups(path, mt, W, lambda p: sorted(p, key=lambda n: [t[0] for t in tab].index(n)), rename=opaque)
    with pytest.raises(ValueError, match='cannot place'):
        kerasfile.load_weights(path, tab, mt)


def test_model_save_weights_keeps_the_format_of_the_extension(tmp_path):
    """the atomic save wrote HDF5 into '<path>.partial.<pid>' and renamed it onto x.npz."""
    from collections import OrderedDict
    for mt in ('tiny_L3',):
        m, _, _ = model.MODELS[mt]()
        rng = np.random.RandomState(11)
        m._host_weights = OrderedDict((n, rng.randn(*s).astype(np.float32)) for n, s, _ in m.param_table())
        W = dict(m._host_weights)
        for name in ('w.npz', 'w.h5'):
            path = str(tmp_path / name)
            m.save_weights(path)
            assert sorted(os.listdir(str(tmp_path))) == sorted(set(os.listdir(str(tmp_path))))
            assert not [f for f in os.listdir(str(tmp_path)) if 'partial' in f]
            m2, _, _ = model.MODELS[mt]()
            m2.load_weights(path)
            assert all(np.array_equal(m2._host_weights[k], W[k]) for k in W), name
        with open(str(tmp_path / 'w.h5'), 'rb') as fh:
            assert fh.read(8) == b'\x89HDF\r\n\x1a\n'
        assert np.load(str(tmp_path / 'w.npz')).files


@pytest.mark.skipif(not _have_h5py(), reason='no h5py interpreter in this container')
def test_h5py_written_interleaved_nested_model_is_placed_by_name(tmp_path):
    """A file written by real libhdf5 whose nested-model weights are layer-interleaved (kernel, bias, gamma, beta,
    moving_mean, moving_variance per layer) -- the order keras would use if a nested model's weights were NOT
    trainable-then-non-trainable: every tensor must land in its named slot."""
    mt = 'cnn_L3_melspec2'
    tab = model._host_param_table(mt)
    W = _random_weights(mt, 9)
    np.savez(str(tmp_path / 'w.npz'), **{k.replace('/', '__'): v for k, v in W.items()})
    order = [n for n, _, _ in tab]
    (tmp_path / 'order.txt').write_text('\n'.join(order))
    script = r'''
import sys, h5py, numpy as np
npz, order_txt, out_path = sys.argv[1:4]
W = {k.replace('__', '/'): v for k, v in np.load(npz).items()}
order = open(order_txt).read().split('\n')
def tf_name(p):
    parts = p.split('/')
    if parts[0] in ('vision_model', 'audio_model'):
        parts = parts[1:]
    if parts[-1] == 'freq2mel':
        parts[-1] = 'Variable'
    return '/'.join(parts) + ':0'
layers = ['input_1', 'input_2', 'vision_model', 'audio_model', 'concatenate_1', 'dense_1', 'dense_2']
out = h5py.File(out_path, 'w')
out.attrs['layer_names'] = [n.encode('utf8') for n in layers]
out.attrs['backend'] = b'tensorflow'
out.attrs['keras_version'] = b'2.0.9'
for name in layers:
    g = out.create_group(name)
    members = [p for p in order if p.split('/')[0] == name]          # ledger order = layer-interleaved
    g.attrs['weight_names'] = [tf_name(p).encode('utf8') for p in members]
    for p in members:
        d = g.create_dataset(tf_name(p), W[p].shape, dtype='float32')
        d[...] = W[p]
out.close()
print('OK')
'''
    path = str(tmp_path / 'h5py_interleaved.h5')
    res = subprocess.run([CONDA_PY, '-c', script, str(tmp_path / 'w.npz'), str(tmp_path / 'order.txt'), path],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    rep = {}
    R = kerasfile.load_weights(path, tab, mt, report=rep)
    assert set(R) == set(W) and all(np.array_equal(R[k], W[k]) for k in W)
    assert rep == {'vision_model': 'name', 'audio_model': 'name', 'dense_1': 'name', 'dense_2': 'name'}
