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
(`Container.weights`); that is the order this module WRITES.  Reading does not rely on it: every
dataset carries its tensorflow variable name (`conv2d_3/kernel:0`, `batch_normalization_2/gamma:0`),
and `load_weights` places tensors by NAME (`_match_group`):

  1. exact names (with the tower prefix or without, kapre's unnamed `Variable:0` for freq2mel) --
     any order inside the group, e.g. layer-interleaved;
  2. if the auto-numbers differ (keras numbers layers by construction order in the session: the
     audio convolutions are conv2d_8..15 only when the vision tower was built first, model.py:280-281),
     by (layer class, rank of the number inside the group, variable name);
  3. only if the names cannot be interpreted at all: by position, and then every file name must
     still END in the variable kind the slot expects (kernel / bias / gamma / beta / moving_mean /
     moving_variance) -- all BatchNorm tensors of a layer share one shape, so a shape check alone
     cannot see a permutation.
The container format is handled by `h5lite` (no h5py needed); legacy `.npz` files are accepted.
"""
import re
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


_LAYER_RE = re.compile(r'^(?P<cls>[A-Za-z0-9_]*?[A-Za-z])_(?P<num>[0-9]+)$')


def _split_name(wname):
    """'vision_model/conv2d_3/kernel:0' -> ('conv2d_3', 'kernel'); None when the name has no such form."""
    n = wname[:-2] if wname.endswith(':0') else wname
    parts = n.split('/')
    if len(parts) < 2:
        return None
    return parts[-2], parts[-1]


def _class_rank(layer_names):
    """{layer: (class, rank)} -- rank of a layer's auto-number among the layers of its class, in ascending order."""
    by_cls = {}
    for ln in set(layer_names):
        m = _LAYER_RE.match(ln)
        if m is None:                       # an explicitly named layer (vision_embedding_layer): a class of its own
            by_cls.setdefault(ln, []).append((0, ln))
        else:
            by_cls.setdefault(m.group('cls'), []).append((int(m.group('num')), ln))
    out = {}
    for cls, items in by_cls.items():
        for rank, (_, ln) in enumerate(sorted(items)):
            out[ln] = (cls, rank)
    return out


def _match_group(gname, pnames, wnames):
    """Which file dataset feeds which parameter of one layer group: -> ([file weight name per pnames entry], mode),
    mode in ('name', 'class-rank', 'position').  See the module docstring."""
    if len(wnames) != len(pnames):
        raise ValueError('Layer "%s" expects %d weights, but the saved weights have %d elements.'
                         % (gname, len(pnames), len(wnames)))
    if len(set(wnames)) != len(wnames):
        raise ValueError('Layer "%s": duplicate weight names in the file' % gname)
    have = set(wnames)
    # 1. exact names
    picked = []
    for p in pnames:
        short = _tf_name(p)
        cands = [short, p.split('/')[0] + '/' + short]
        if p.endswith('/freq2mel'):
            cands += [c.replace('/Variable:0', '/freq2mel:0') for c in cands]
        hit = [c for c in cands if c in have]
        if len(hit) != 1:
            picked = None
            break
        picked.append(hit[0])
    if picked is not None and len(set(picked)) == len(picked):
        return picked, 'name'
    # 2. (class, rank of the auto-number, variable)
    fsplit = [_split_name(w) for w in wnames]
    psplit = [_split_name(_tf_name(p)) for p in pnames]
    if all(x is not None for x in fsplit):
        frank = _class_rank([x[0] for x in fsplit])
        prank = _class_rank([x[0] for x in psplit])
        if frank is not None and prank is not None:
            alias = lambda v: 'freq2mel' if v == 'Variable' else v            # noqa: E731
            index = {}
            for w, (ln, var) in zip(wnames, fsplit):
                index[(frank[ln], alias(var))] = w
            picked = [index.get((prank[ln], alias(var))) for ln, var in psplit]
            if len(index) == len(wnames) and all(x is not None for x in picked) and len(set(picked)) == len(picked):
                return picked, 'class-rank'
    # 3. position, guarded by the variable kind
    for w, p in zip(wnames, pnames):
        kind = p.rsplit('/', 1)[1]
        fs = _split_name(w)
        fkind = fs[1] if fs is not None else (w[:-2] if w.endswith(':0') else w)
        if kind == 'freq2mel' and fkind == 'Variable':
            continue
        if fkind != kind:
            raise ValueError('Layer "%s": cannot place "%s" by name, and by position it would land in the %s slot "%s"'
                             % (gname, w, kind, p))
    return list(wnames), 'position'


def load_weights(path, table, model_type, wrapper=False, report=None):
    """-> OrderedDict param name -> float32 array.  Tensors are placed by dataset NAME, whatever their order inside the
    layer group (module docstring); `report`, when given, receives {layer group: match mode}."""
    if path.endswith('.npz'):
        z = np.load(path)
        return OrderedDict((k.replace('__', '/'), z[k]) for k in z.files)
    root = h5lite.read_file(path)
    if 'layer_names' not in root.attrs:
        raise ValueError('%s is not a keras weight file (no layer_names attribute)' % path)
    layer_names = [n.decode('utf8') if isinstance(n, bytes) else str(n) for n in np.atleast_1d(root.attrs['layer_names'])]
    shapes = {n: tuple(s) for n, s, _ in table}
    expected = OrderedDict((g, p) for g, p in keras_groups(table, model_type, wrapper).items() if p)
    file_groups = OrderedDict()
    for ln in layer_names:
        g = root.children.get(ln)
        if g is None:
            raise ValueError('layer group "%s" missing in %s' % (ln, path))
        wn = g.attrs.get('weight_names')
        wn = [] if wn is None or np.asarray(wn).dtype.kind != 'S' else [n.decode('utf8') for n in np.atleast_1d(wn)]
        if wn:
            file_groups[ln] = (g, wn)
    if len(file_groups) != len(expected):
        raise ValueError('You are trying to load a weight file containing %d layers into a model with %d layers.'
                         % (len(file_groups), len(expected)))
    # layer groups by name when the file has them all (any order), else in file order like keras
    if set(file_groups) == set(expected):
        pairs = [(gname, file_groups[gname], pnames) for gname, pnames in expected.items()]
    else:
        pairs = [(gname, fg, pnames) for fg, (gname, pnames) in zip(file_groups.values(), expected.items())]
    out = OrderedDict()
    for gname, (g, wn), pnames in pairs:
        picked, mode = _match_group(gname, pnames, wn)
        if report is not None:
            report[gname] = mode
        for w, p in zip(picked, pnames):
            arr = np.asarray(g[w], dtype=np.float32)
            if tuple(arr.shape) != shapes[p]:
                raise ValueError('shape mismatch for %s: file %s has %s, model expects %s' % (p, w, arr.shape, shapes[p]))
            out[p] = arr
    return out
