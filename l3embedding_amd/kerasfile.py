"""Keras-2.0.9 HDF5 weight files (model.save_weights / load_weights) -- placeholder that
is replaced by the self-contained HDF5 implementation in h5lite.py."""
import numpy as np


def save_weights(path, named, table, model_type, wrapper=False):
    np.savez(path if path.endswith('.npz') else path + '.npz', **{k.replace('/', '__'): v for k, v in named.items()})


def load_weights(path, table, model_type, wrapper=False):
    z = np.load(path if path.endswith('.npz') else path + '.npz')
    return {k.replace('__', '/'): z[k] for k in z.files}
