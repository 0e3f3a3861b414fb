"""CPU tests of the host-side train.py mirror: HDF5 batch-blob reader (train.py:142-205
semantics), preprocessing, callbacks and run-directory bookkeeping."""
import os
import random

import numpy as np
import pytest

from l3embedding_amd import h5lite, train as T


def _write_blobs(d, n_files=3, per=5, seed=0):
    rng = np.random.RandomState(seed)
    blobs = {}
    for i in range(n_files):
        a = rng.randint(-32768, 32768, (per, 1, 48000)).astype(np.int16)
        v = rng.randint(0, 256, (per, 224, 224, 3)).astype(np.uint8)
        lab = rng.randint(0, 2, per)
        l = np.stack([lab, 1 - lab], 1).astype(np.int64)
        root = h5lite.Group()
        root.create_dataset('audio', a, compression='gzip')          # data/avc/sample.py:565-568
        root.create_dataset('video', v, compression='gzip')
        root.create_dataset('label', l, compression='gzip')
        root.create_dataset('audio_start_sample_idx', np.arange(per))
        name = '%d_%d_%d.h5' % (20171021 + i, i, 0)
        h5lite.write_file(os.path.join(d, name), root)
        blobs[name] = (a, v, l)
    return blobs


def test_data_generator_matches_reference_semantics(tmp_path):
    d = str(tmp_path)
    blobs = _write_blobs(d)
    order = os.listdir(d)
    A = np.concatenate([blobs[f][0] for f in order])
    V = np.concatenate([blobs[f][1] for f in order])
    L = np.concatenate([blobs[f][2] for f in order])
    gen = T.data_generator(d, batch_size=4, random_state=7)
    b0, b1, b2 = next(gen), next(gen), next(gen)
    assert set(b0) == {'audio', 'video', 'label'}               # metadata keys dropped (train.py:150-152)
    assert b0['audio'].dtype == np.float32 and b0['video'].dtype == np.float32
    assert np.array_equal(b1['label'], L[4:8])                  # batches span blob boundaries (train.py:161-176)
    assert np.array_equal(b0['audio'], (A[0:4].astype(np.float32) / 32768))
    assert np.array_equal(b2['video'], (2 * (V[8:12].astype(np.float64) / 255).astype(np.float32) - 1))
    assert b0['video'].min() >= -1 and b0['video'].max() <= 1
    # resume: skipping the first 2 batches yields the third (train.py:164-193)
    gen2 = T.data_generator(d, batch_size=4, random_state=7, start_batch_idx=2)
    b = next(gen2)
    assert np.array_equal(b['label'], b2['label']) and np.array_equal(b['audio'], b2['audio'])
    # raw mode keeps the stored integer tensors for on-GPU scaling
    r = next(T.data_generator(d, batch_size=4, random_state=7, raw=True))
    assert r['audio'].dtype == np.int16 and r['video'].dtype == np.uint8
    x, y = next(T.keras_tuples(T.data_generator(d, batch_size=2), ['video', 'audio'], 'label'))
    assert len(x) == 2 and x[0].shape == (2, 224, 224, 3) and x[1].shape == (2, 1, 48000) and y.shape == (2, 2)
    # after one pass the file list is reshuffled with the seeded RNG and the stream continues
    gen3 = T.data_generator(d, batch_size=5, random_state=1)
    seen = [next(gen3)['label'] for _ in range(6)]
    assert all(s.shape == (5, 2) for s in seen)
    se = T.single_epoch_data_generator(d, 2, batch_size=5, random_state=1)
    e = [next(se)['label'] for _ in range(4)]
    assert np.array_equal(e[0], e[2]) and np.array_equal(e[1], e[3])      # restarts every epoch_size batches


def test_pcm2float_and_errors():
    assert T.pcm2float(np.array([-32768, 0, 32767], np.int16), 'float32').tolist() == [-1.0, 0.0, 32767 / 32768]
    with pytest.raises(TypeError):
        T.pcm2float(np.zeros(2, np.float32))
    with pytest.raises(TypeError):
        T.pcm2float(np.zeros(2, np.int16), 'int32')


class _FakeModel(object):
    def __init__(self):
        self.saved = []

    def save_weights(self, path, overwrite=True):
        self.saved.append(os.path.basename(path))


def test_checkpoint_and_csv_callbacks(tmp_path):
    m = _FakeModel()
    latest = T.ModelCheckpoint(str(tmp_path / 'model_latest.h5'), save_weights_only=True)
    best_acc = T.ModelCheckpoint(str(tmp_path / 'model_best_valid_accuracy.h5'), save_best_only=True, monitor='val_acc')
    best_loss = T.ModelCheckpoint(str(tmp_path / 'model_best_valid_loss.h5'), save_best_only=True, monitor='val_loss')
    every = T.ModelCheckpoint(str(tmp_path / 'model_checkpoint.{epoch:02d}.h5'), period=2)
    csvl = T.CSVLogger(str(tmp_path / 'history_csvlog.csv'), append=True)
    lh = T.LossHistory(str(tmp_path / 'history_checkpoint.pkl'))
    cbs = [latest, best_acc, best_loss, every, csvl, lh]
    for c in cbs:
        c.set_model(m)
        c.on_train_begin({})
    logs = [dict(loss=1.0, acc=0.5, val_loss=0.9, val_acc=0.55), dict(loss=0.8, acc=0.6, val_loss=1.1, val_acc=0.60),
            dict(loss=0.7, acc=0.7, val_loss=0.8, val_acc=0.58)]
    for ep, lg in enumerate(logs):
        for c in cbs:
            c.on_epoch_end(ep, lg)
    for c in cbs:
        c.on_train_end({})
    assert m.saved.count('model_latest.h5') == 3
    assert m.saved.count('model_best_valid_accuracy.h5') == 2          # 0.55 -> 0.60, not 0.58
    assert m.saved.count('model_best_valid_loss.h5') == 2              # 0.9, then 0.8
    assert 'model_checkpoint.02.h5' in m.saved and 'model_checkpoint.01.h5' not in m.saved
    rows = open(str(tmp_path / 'history_csvlog.csv')).read().strip().split('\n')
    assert rows[0] == 'epoch,acc,loss,val_acc,val_loss' and len(rows) == 4   # 04_plot_training_history.py:32-36
    assert T.get_restart_info(str(tmp_path / 'history_csvlog.csv')) == (2, 0.58, 0.8)
    # resuming appends without a second header (train.py:363-365 append=True)
    csv2 = T.CSVLogger(str(tmp_path / 'history_csvlog.csv'), append=True)
    csv2.set_model(m)
    csv2.on_train_begin({})
    csv2.on_epoch_end(3, logs[0])
    csv2.on_train_end({})
    rows = open(str(tmp_path / 'history_csvlog.csv')).read().strip().split('\n')
    assert len(rows) == 5 and rows.count('epoch,acc,loss,val_acc,val_loss') == 1
