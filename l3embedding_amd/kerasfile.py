"""Keras-2.0.9 HDF5 weight files for the L3 models.

Writes/reads the layout `Model.save_weights` / `load_weights` of keras 2.0.9 use
(reference call sites: l3embedding/train.py:316-355 ModelCheckpoint(save_weights_only=True),
l3embedding/model.py:119 `m.load_weights(path)`, 05_generate_embedding_samples.py:154-157):

  /            attrs layer_names (fixed-length byte strings), backend, keras_version
  /<layer>     one group per top-level layer, attr weight_names; weights live at
               /<layer>/<weight_name>, e.g. /vision_model/conv2d_1/kernel:0
  single-GPU model  layers: input_1, input_2, vision_model, audio_model, concatenate_1, dense_1, dense_2
  multi-GPU wrapper (training_utils.multi_gpu_model) one weighted layer named like the
               template model (model.py:77,117-119) holding every weight

[3P] keras orders a nested model's weights as trainable-then-non-trainable
(`Container.weights`) and `load_weights` is positional inside each layer group; reading here
is therefore positional too (names are only informative: kapre's freq2mel variable is
unnamed in kapre 0.1.3.1 and shows up as `melspectrogram_1/Variable:0`).
The container format is handled by `h5lite` (no h5py needed); legacy `.npz` files are accepted.
"""
from collections import OrderedDict

import numpy as np

from . import h5lite


def _tf_name(pname):
    """'vision_model/conv2d_1/kernel' -> 'conv2d_1/kernel:0' ; 'dense_1/bias' -> 'dense_1/bias:0'"""
    parts = pname.split('/')
    if parts[0] in ('vision_model', 'audio_model'):
        parts = parts[1:]
    if parts[-1] == 'freq2mel':
        parts[-1] = 'Variable'
    return '/'.join(parts) + ':0'


def keras_groups(table, model_type, wrapper=False):
    """OrderedDict layer-group -> [param names in keras `layer.weights` order]."""
    groups = OrderedDict()

    def nested(prefix):
        tr = [n for n, _, t in table if n.startswith(prefix + '/') and t]
        nt = [n for n, _, t in table if n.startswith(prefix + '/') and not t]
        return tr + nt

    if not wrapper:
        groups['input_1'] = []
        groups['input_2'] = []
        groups['vision_model'] = nested('vision_model')
        groups['audio_model'] = nested('audio_model')
        groups['concatenate_1'] = []
        groups['dense_1'] = [n for n, _, _ in table if n.startswith('dense_1/')]
        groups['dense_2'] = [n for n, _, _ in table if n.startswith('dense_2/')]
    else:
        tr = [n for n, _, t in table if t]
        nt = [n for n, _, t in table if not t]
        groups['input_1'] = []
        groups['input_2'] = []
        groups[model_type] = tr + nt
        groups['concatenate_2'] = []
    return groups


def save_weights(path, named, table, model_type, wrapper=False):
    if path.endswith('.npz'):
        np.savez(path, **{k.replace('/', '__'): v for k, v in named.items()})
        return
    root = h5lite.Group()
    groups = keras_groups(table, model_type, wrapper)
    root.attrs['layer_names'] = np.array([g.encode('utf8') for g in groups])
    root.attrs['backend'] = b'tensorflow'
    root.attrs['keras_version'] = b'2.0.9'
    for gname, pnames in groups.items():
        g = root.create_group(gname)
        wnames = [_tf_name(p) for p in pnames]
        g.attrs['weight_names'] = np.array([w.encode('utf8') for w in wnames]) if wnames else np.zeros((0,), np.float64)
        for p, w in zip(pnames, wnames):
            g.create_dataset(w, np.asarray(named[p], dtype=np.float32))
    h5lite.write_file(path, root)


def load_weights(path, table, model_type, wrapper=False):
    """-> OrderedDict param name -> float32 array (positional per layer group, like keras)."""
    if path.endswith('.npz'):
        z = np.load(path)
        return OrderedDict((k.replace('__', '/'), z[k]) for k in z.files)
    root = h5lite.read_file(path)
    if 'layer_names' not in root.attrs:
        raise ValueError('%s is not a keras weight file (no layer_names attribute)' % path)
    layer_names = [n.decode('utf8') if isinstance(n, bytes) else str(n) for n in np.atleast_1d(root.attrs['layer_names'])]
    shapes = {n: tuple(s) for n, s, _ in table}
    expected = OrderedDict((g, p) for g, p in keras_groups(table, model_type, wrapper).items() if p)
    file_groups = []
    for ln in layer_names:
        g = root.children.get(ln)
        if g is None:
            raise ValueError('layer group "%s" missing in %s' % (ln, path))
        wn = g.attrs.get('weight_names')
        wn = [] if wn is None or np.asarray(wn).dtype.kind != 'S' else [n.decode('utf8') for n in np.atleast_1d(wn)]
        if wn:
            file_groups.append((ln, g, wn))
    if len(file_groups) != len(expected):
        raise ValueError('You are trying to load a weight file containing %d layers into a model with %d layers.'
                         % (len(file_groups), len(expected)))
    out = OrderedDict()
    for (ln, g, wn), (gname, pnames) in zip(file_groups, expected.items()):
        if len(wn) != len(pnames):
            raise ValueError('Layer "%s" expects %d weights, but the saved weights have %d elements.'
                             % (gname, len(pnames), len(wn)))
        for w, p in zip(wn, pnames):
            arr = np.asarray(g[w], dtype=np.float32)
            if tuple(arr.shape) != shapes[p]:
                raise ValueError('shape mismatch for %s: file %s has %s, model expects %s' % (p, w, arr.shape, shapes[p]))
            out[p] = arr
    return out
