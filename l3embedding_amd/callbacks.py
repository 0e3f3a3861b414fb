"""Training-loop callbacks for `L3Model.fit_generator`.

The reference wires five Keras callbacks into its run (l3embedding/train.py:316-365): three
`ModelCheckpoint`s (latest / best val_acc / best val_loss / every N epochs), `CSVLogger`, and its own
`LossHistory` / `TimeHistory` (train.py:29-53,108-131).  Keras is not available on the GPU box, so the
slice of the callback protocol those use is implemented here; what must stay compatible are the
*artefacts* (file names, CSV columns, pickle contents), because `get_restart_info` and
04_plot_training_history.py read them back.
"""
import csv
import os
import pickle
import tempfile
import time

import numpy as np


class Callback(object):
    """Hook points `fit_generator` calls; all default to no-ops."""

    model = None
    params = None

    def set_model(self, model):
        self.model = model

    def set_params(self, params):
        self.params = params

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_batch_begin(self, batch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass


def _replace_file(path, writer, mode='wb'):
    """Write-then-rename so a killed run never leaves a truncated artefact behind."""
    fd, tmp = tempfile.mkstemp(dir=os.path.dirname(os.path.abspath(path)), prefix='.tmp-')
    try:
        with os.fdopen(fd, mode) as fh:
            writer(fh)
        os.replace(tmp, path)
    except BaseException:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise


class LossHistory(Callback):
    """`history_checkpoint.pkl`: {'loss': [...], 'val_loss': [...]}, one entry per finished epoch,
    rewritten after every epoch (artefact of train.py:29-53)."""

    SERIES = ('loss', 'val_loss')

    def __init__(self, outfile):
        self.outfile = outfile
        self._series = {k: [] for k in self.SERIES}

    loss = property(lambda self: self._series['loss'])
    val_loss = property(lambda self: self._series['val_loss'])

    def on_train_begin(self, logs=None):
        self._series = {k: [] for k in self.SERIES}

    def on_epoch_end(self, epoch, logs=None):
        for k in self.SERIES:
            self._series[k].append((logs or {}).get(k))
        snapshot = {k: list(v) for k, v in self._series.items()}
        _replace_file(self.outfile, lambda fh: pickle.dump(snapshot, fh))


class TimeHistory(Callback):
    """Wall-clock seconds per epoch and per batch, kept in `epoch_times` / `batch_times` (train.py:108-131).

    Its batch hooks only read the clock, so fit_generator may keep its pipelined order (step k + 1 is enqueued before step k's
    results are read): `batch_times` then holds the host's turn-around per step -- which in steady state IS the step time, the
    device being the bottleneck -- not "launch to completion" of one step."""

    batch_hooks_are_passive = True

    def __init__(self, logger=None):
        self._log = logger
        self.epoch_times, self.batch_times = [], []
        self._t = {}

    def on_train_begin(self, logs=None):
        self.epoch_times, self.batch_times = [], []

    def _tick(self, kind):
        self._t[kind] = time.perf_counter()

    def _tock(self, kind, sink, level):
        took = time.perf_counter() - self._t.pop(kind, time.perf_counter())
        sink.append(took)
        if self._log is not None:
            getattr(self._log, level)('%s took %s seconds', kind, took)

    def on_epoch_begin(self, epoch, logs=None):
        self._tick('Epoch')

    def on_epoch_end(self, epoch, logs=None):
        self._tock('Epoch', self.epoch_times, 'info')

    def on_batch_begin(self, batch, logs=None):
        self._tick('Batch')

    def on_batch_end(self, batch, logs=None):
        self._tock('Batch', self.batch_times, 'debug')


class ModelCheckpoint(Callback):
    """[3P] keras.callbacks.ModelCheckpoint as the reference configures it (train.py:329-355): weights
    only; optionally only when `monitor` improves; every `period` epochs; `{epoch:02d}` in the file name
    is the 1-based epoch -- the reference's own artefacts say so: a default 150-epoch, checkpoint-interval-10
    run left `model_checkpoint.150.h5` (notebooks/extract_spectrogram_models_from_avc_models.ipynb:105,
    test_load_converted_model.ipynb:77), which a 0-based `{epoch}` (saves at 9, 19, ... 149) cannot produce.  `best` and `epochs_since_last_save` are public because a resumed run seeds
    them (train.py:333-334,342-343,352-353)."""

    def __init__(self, filepath, monitor='val_loss', verbose=0, save_best_only=False, save_weights_only=True,
                 mode='auto', period=1, logger=None):
        if mode not in ('auto', 'min', 'max'):
            mode = 'auto'
        higher_is_better = mode == 'max' or (mode == 'auto' and ('acc' in monitor or monitor.startswith('fmeasure')))
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.save_weights_only, self.period = save_best_only, save_weights_only, int(period)
        self._improved = (lambda new, old: new > old) if higher_is_better else (lambda new, old: new < old)
        self.best = -np.inf if higher_is_better else np.inf
        self.epochs_since_last_save = 0
        self._log = logger

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.epochs_since_last_save += 1
        if self.epochs_since_last_save < self.period:
            return
        self.epochs_since_last_save = 0
        if self.save_best_only:
            value = logs.get(self.monitor)
            if value is None or not self._improved(value, self.best):
                return
            self.best = value
        target = self.filepath.format(epoch=epoch + 1, **logs)
        if self.verbose and self._log is not None:
            self._log.info('Epoch %05d: saving model to %s', epoch + 1, target)
        self.model.save_weights(target, overwrite=True)


class CSVLogger(Callback):
    """[3P] keras.callbacks.CSVLogger(append=True): header `epoch` + the sorted log keys (so
    `epoch,acc,loss,val_acc,val_loss`), one row per epoch with the 0-based epoch index; a resumed run
    appends without repeating the header.  Read back by `get_restart_info` (train.py:208-215)."""

    def __init__(self, filename, separator=',', append=False):
        self.filename, self.sep, self.append = filename, separator, append
        self._fh = self._columns = None

    def on_train_begin(self, logs=None):
        resume = self.append and os.path.exists(self.filename) and os.path.getsize(self.filename) > 0
        self._need_header = not resume
        self._fh = open(self.filename, 'a' if self.append else 'w', newline='')

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        if self._columns is None:
            self._columns = ['epoch'] + sorted(logs)
        out = csv.writer(self._fh, delimiter=self.sep)
        if self._need_header:
            out.writerow(self._columns)
            self._need_header = False
        out.writerow([epoch] + [logs.get(k, 'NA') for k in self._columns[1:]])
        self._fh.flush()

    def on_train_end(self, logs=None):
        if self._fh is not None:
            self._fh.close()
            self._fh = None


def last_epoch_record(history_path):
    """(epoch index, val_acc, val_loss) of the last row of a `history_csvlog.csv`."""
    with open(history_path, newline='') as fh:
        rows = list(csv.reader(fh))
    if len(rows) < 2:
        raise ValueError('no finished epoch in "{}"'.format(history_path))
    rec = dict(zip(rows[0], rows[-1]))
    return int(rec['epoch']), float(rec['val_acc']), float(rec['val_loss'])
