"""Host-side mirror of l3embedding/train.py on top of the MI355X engine.

Same function names, argument meaning and run-directory artefacts as the reference:

  cycle_shuffle / data_generator / single_epoch_data_generator   train.py:134-205
  get_restart_info                                                train.py:208-215
  LossHistory / TimeHistory                                       train.py:29-53,108-131
  train(...)                                                      train.py:218-421
  ModelCheckpoint / CSVLogger (keras callbacks the reference instantiates, train.py:316-365)

Out of scope here: Google-Sheets logging, GitPython commit stamping,
the rotating file logger.  HDF5 batch blobs (data/avc/sample.py:565-568, gzip) are read with
the self-contained `h5lite`; with `raw=True` the generator yields the stored uint8/int16
tensors so the scaling of train.py:186,189 runs on the GPU (`l3_upload_batch_raw`).
"""
import csv
import datetime
import getpass
import json
import logging
import os
import pickle
import random
import time

import numpy as np

from . import h5lite
from .model import MODELS, Adam, load_model

LOGGER = logging.getLogger('l3embedding')
LOGGER.setLevel(logging.DEBUG)


def pcm2float(sig, dtype='float64'):
    """l3embedding/audio.py:4-31"""
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


def img_as_float(img):
    """[3P] skimage.img_as_float for uint8 input: x / 255 in float64 (train.py:186)."""
    img = np.asarray(img)
    if img.dtype != np.uint8:
        raise TypeError('expected uint8 frames')
    return img.astype(np.float64) / 255.0


# ---------------------------------------------------------------------------------------------
# callbacks
# ---------------------------------------------------------------------------------------------
class Callback(object):
    def __init__(self):
        self.model = None
        self.params = {}

    def set_model(self, model):
        self.model = model

    def set_params(self, params):
        self.params = params

    def on_train_begin(self, logs=None): pass
    def on_train_end(self, logs=None): pass
    def on_epoch_begin(self, epoch, logs=None): pass
    def on_epoch_end(self, epoch, logs=None): pass
    def on_batch_begin(self, batch, logs=None): pass
    def on_batch_end(self, batch, logs=None): pass


class LossHistory(Callback):
    """train.py:29-53"""

    def __init__(self, outfile):
        super().__init__()
        self.outfile = outfile

    def on_train_begin(self, logs=None):
        self.loss = []
        self.val_loss = []

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.loss.append(logs.get('loss'))
        self.val_loss.append(logs.get('val_loss'))
        with open(self.outfile, 'wb') as fp:
            pickle.dump({'loss': self.loss, 'val_loss': self.val_loss}, fp)


class TimeHistory(Callback):
    """train.py:108-131"""

    def on_train_begin(self, logs=None):
        self.epoch_times = []
        self.batch_times = []

    def on_epoch_begin(self, epoch, logs=None):
        self.epoch_start = time.time()

    def on_epoch_end(self, epoch, logs=None):
        t = time.time() - self.epoch_start
        LOGGER.info('Epoch took {} seconds'.format(t))
        self.epoch_times.append(t)

    def on_batch_begin(self, batch, logs=None):
        self.batch_start = time.time()

    def on_batch_end(self, batch, logs=None):
        t = time.time() - self.batch_start
        LOGGER.debug('Batch took {} seconds'.format(t))
        self.batch_times.append(t)


class ModelCheckpoint(Callback):
    """[3P] keras.callbacks.ModelCheckpoint as configured at train.py:329-355
    (save_weights_only, save_best_only, monitor val_acc/val_loss, period)."""

    def __init__(self, filepath, monitor='val_loss', verbose=0, save_best_only=False, save_weights_only=False,
                 mode='auto', period=1):
        super().__init__()
        self.filepath = filepath
        self.monitor = monitor
        self.verbose = verbose
        self.save_best_only = save_best_only
        self.save_weights_only = save_weights_only
        self.period = period
        self.epochs_since_last_save = 0
        if mode == 'max' or (mode == 'auto' and ('acc' in monitor or monitor.startswith('fmeasure'))):
            self.monitor_op, self.best = np.greater, -np.inf
        else:
            self.monitor_op, self.best = np.less, np.inf

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.epochs_since_last_save += 1
        if self.epochs_since_last_save < self.period:
            return
        self.epochs_since_last_save = 0
        filepath = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            current = logs.get(self.monitor)
            if current is None or not self.monitor_op(current, self.best):
                return
            self.best = current
        if self.verbose:
            LOGGER.info('Epoch %05d: saving model to %s' % (epoch + 1, filepath))
        self.model.save_weights(filepath, overwrite=True)


class CSVLogger(Callback):
    """[3P] keras.callbacks.CSVLogger(append=True): columns epoch + sorted(log keys), i.e.
    epoch,acc,loss,val_acc,val_loss as read back by get_restart_info and 04_plot_training_history.py:32-36."""

    def __init__(self, filename, separator=',', append=False):
        super().__init__()
        self.filename = filename
        self.sep = separator
        self.append = append
        self.keys = None

    def on_train_begin(self, logs=None):
        self.append_header = not (self.append and os.path.exists(self.filename) and os.path.getsize(self.filename) > 0)
        self.csv_file = open(self.filename, 'a' if self.append else 'w', newline='')
        self.writer = None

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        if self.keys is None:
            self.keys = sorted(logs.keys())
        if self.writer is None:
            self.writer = csv.DictWriter(self.csv_file, fieldnames=['epoch'] + self.keys, delimiter=self.sep)
            if self.append_header:
                self.writer.writeheader()
        row = {'epoch': epoch}
        row.update((k, logs.get(k, 'NA')) for k in self.keys)
        self.writer.writerow(row)
        self.csv_file.flush()

    def on_train_end(self, logs=None):
        self.csv_file.close()
        self.writer = None


# ---------------------------------------------------------------------------------------------
# data feed (train.py:134-205)
# ---------------------------------------------------------------------------------------------
def cycle_shuffle(iterable, shuffle=True):
    lst = list(iterable)
    while True:
        yield from lst
        if shuffle:
            random.shuffle(lst)


def _read_blob(path, keys):
    root = h5lite.read_file(path)
    return {k: root[k] for k in keys}


def data_generator(data_dir, batch_size=512, random_state=20180123, start_batch_idx=None, keys=None, raw=False):
    """Yields {'video','audio','label'} batches assembled across HDF5 blobs exactly like
    train.py:142-195 (same file order, spill-over and skip logic)."""
    random.seed(random_state)
    batch = None
    curr_batch_size = 0
    batch_idx = 0
    if not keys:
        keys = ['audio', 'video', 'label']
    for fname in cycle_shuffle(os.listdir(data_dir)):
        batch_path = os.path.join(data_dir, fname)
        blob_start_idx = 0
        blob = _read_blob(batch_path, keys)
        blob_size = len(blob['label'])
        while blob_start_idx < blob_size:
            blob_end_idx = min(blob_start_idx + batch_size - curr_batch_size, blob_size)
            if start_batch_idx is None or batch_idx >= start_batch_idx:
                if batch is None:
                    batch = {k: blob[k][blob_start_idx:blob_end_idx] for k in keys}
                else:
                    for k in keys:
                        batch[k] = np.concatenate([batch[k], blob[k][blob_start_idx:blob_end_idx]])
            curr_batch_size += blob_end_idx - blob_start_idx
            blob_start_idx = blob_end_idx
            if curr_batch_size == batch_size:
                if start_batch_idx is None or batch_idx >= start_batch_idx:
                    if not raw:
                        batch['video'] = 2 * img_as_float(batch['video']).astype('float32') - 1
                        batch['audio'] = pcm2float(batch['audio'], dtype='float32')
                    yield batch
                batch_idx += 1
                curr_batch_size = 0
                batch = None


def single_epoch_data_generator(data_dir, epoch_size, **kwargs):
    while True:
        data_gen = data_generator(data_dir, **kwargs)
        for idx, item in enumerate(data_gen):
            yield item
            if (idx + 1) == epoch_size:
                break


def keras_tuples(stream, inputs, outputs):
    """[3P] pescador.maps.keras_tuples as used at train.py:382-384,393-395."""
    for data in stream:
        x = [data[k] for k in inputs] if isinstance(inputs, (list, tuple)) else data[inputs]
        y = data[outputs]
        yield (x, y)


def get_restart_info(history_path):
    last = None
    with open(history_path, 'r') as f:
        reader = csv.DictReader(f)
        for row in reader:
            last = row
    return int(last['epoch']), float(last['val_acc']), float(last['val_loss'])


# ---------------------------------------------------------------------------------------------
# train()  (train.py:218-421)
# ---------------------------------------------------------------------------------------------
def train(train_data_dir, validation_data_dir, output_dir, num_epochs=300, train_epoch_size=4096,
          validation_epoch_size=1024, train_batch_size=64, validation_batch_size=64, model_type='cnn_L3_orig',
          random_state=20180123, learning_rate=1e-4, verbose=False, checkpoint_interval=10, log_path=None,
          disable_logging=False, gpus=1, continue_model_dir=None, gsheet_id=None, google_dev_app_name=None):
    if not disable_logging and log_path:
        fh = logging.FileHandler(log_path)
        fh.setFormatter(logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s'))
        LOGGER.addHandler(fh)
    model_id = os.path.basename(os.path.normpath(train_data_dir))
    param_dict = {
        'username': getpass.getuser(), 'train_data_dir': train_data_dir, 'validation_data_dir': validation_data_dir,
        'model_id': model_id, 'output_dir': output_dir, 'num_epochs': num_epochs, 'train_epoch_size': train_epoch_size,
        'validation_epoch_size': validation_epoch_size, 'train_batch_size': train_batch_size,
        'validation_batch_size': validation_batch_size, 'model_type': model_type, 'random_state': random_state,
        'learning_rate': learning_rate, 'verbose': verbose, 'checkpoint_interval': checkpoint_interval, 'gpus': gpus,
        'continue_model_dir': continue_model_dir, 'backend': 'libl3hip (MI355X)',
    }
    LOGGER.info('Training with the following arguments: {}'.format(param_dict))
    if gsheet_id:
        LOGGER.warning('Google-Sheets logging is out of scope for this build; ignoring gsheet_id')

    if continue_model_dir:
        latest_model_path = os.path.join(continue_model_dir, 'model_latest.h5')
        m, inputs, outputs = load_model(latest_model_path, model_type, return_io=True, src_num_gpus=gpus)
    else:
        m, inputs, outputs = MODELS[model_type](num_gpus=gpus)

    loss = 'categorical_crossentropy'
    metrics = ['accuracy']
    if continue_model_dir:
        model_dir = continue_model_dir
    else:
        model_dir = os.path.join(output_dir, 'embedding', model_id, datetime.datetime.now().strftime("%Y%m%d%H%M%S"))
    # one process per GPU: rank 0 owns the run directory and every file written into it
    is_main = True
    if gpus > 1:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            is_main = dist.get_rank() == 0
            box = [model_dir]
            dist.broadcast_object_list(box, src=0)
            model_dir = box[0]
    if is_main and not os.path.isdir(model_dir):
        os.makedirs(model_dir)

    LOGGER.info('Compiling model...')
    m.compile(Adam(lr=learning_rate), loss=loss, metrics=metrics)
    LOGGER.info('Model files can be found in "{}"'.format(model_dir))

    param_dict['model_dir'] = model_dir
    if is_main:
        with open(os.path.join(model_dir, 'config.json'), 'w') as fd:
            json.dump(param_dict, fd, indent=2)
        with open(os.path.join(model_dir, 'model_spec.pkl'), 'wb') as fd:
            pickle.dump(m.get_config(), fd)
        with open(os.path.join(model_dir, 'model.json'), 'w') as fd:
            json.dump(m.to_json(), fd, indent=2)

    latest_weight_path = os.path.join(model_dir, 'model_latest.h5')
    best_valid_acc_weight_path = os.path.join(model_dir, 'model_best_valid_accuracy.h5')
    best_valid_loss_weight_path = os.path.join(model_dir, 'model_best_valid_loss.h5')
    checkpoint_weight_path = os.path.join(model_dir, 'model_checkpoint.{epoch:02d}.h5')

    if continue_model_dir is not None:
        prev_train_hist_path = os.path.join(continue_model_dir, 'history_csvlog.csv')
        last_epoch_idx, last_val_acc, last_val_loss = get_restart_info(prev_train_hist_path)

    cb = [ModelCheckpoint(latest_weight_path, save_weights_only=True, verbose=1)]
    best_val_acc_cb = ModelCheckpoint(best_valid_acc_weight_path, save_weights_only=True, save_best_only=True,
                                      verbose=1, monitor='val_acc')
    if continue_model_dir is not None:
        best_val_acc_cb.best = last_val_acc
    cb.append(best_val_acc_cb)
    best_val_loss_cb = ModelCheckpoint(best_valid_loss_weight_path, save_weights_only=True, save_best_only=True,
                                       verbose=1, monitor='val_loss')
    if continue_model_dir is not None:
        best_val_loss_cb.best = last_val_loss
    cb.append(best_val_loss_cb)
    checkpoint_cb = ModelCheckpoint(checkpoint_weight_path, save_weights_only=True, period=checkpoint_interval)
    if continue_model_dir is not None:
        checkpoint_cb.epochs_since_last_save = (last_epoch_idx + 1) % checkpoint_interval
    cb.append(checkpoint_cb)
    cb.append(TimeHistory())
    cb.append(LossHistory(os.path.join(model_dir, 'history_checkpoint.pkl')))
    cb.append(CSVLogger(os.path.join(model_dir, 'history_csvlog.csv'), append=True, separator=','))
    if not is_main:
        cb = [c for c in cb if isinstance(c, TimeHistory)]     # identical weights on every rank: rank 0 writes

    LOGGER.info('Setting up train data generator...')
    train_start_batch_idx = train_epoch_size * (last_epoch_idx + 1) if continue_model_dir is not None else None
    # raw=True: the generator hands over the stored uint8 / int16 arrays and the GPU applies the scalings of
    # train.py:186,189 (bit-exact, `preprocess_*` kernels) -- 3.2x fewer host->device bytes and no float
    # conversion on the single loader thread
    train_gen = keras_tuples(data_generator(train_data_dir, batch_size=train_batch_size, random_state=random_state,
                                            start_batch_idx=train_start_batch_idx, raw=True), ['video', 'audio'], 'label')
    LOGGER.info('Setting up validation data generator...')
    val_gen = keras_tuples(single_epoch_data_generator(validation_data_dir, validation_epoch_size,
                                                       batch_size=validation_batch_size, random_state=random_state,
                                                       raw=True),
                           ['video', 'audio'], 'label')

    LOGGER.info('Fitting model...')
    verbosity = 1 if verbose else 2
    initial_epoch = last_epoch_idx + 1 if continue_model_dir is not None else 0
    history = m.fit_generator(train_gen, train_epoch_size, num_epochs, validation_data=val_gen,
                              validation_steps=validation_epoch_size, callbacks=cb, verbose=verbosity,
                              initial_epoch=initial_epoch)

    LOGGER.info('Done training. Saving results to disk...')
    if is_main:
        with open(os.path.join(model_dir, 'history.pkl'), 'wb') as fd:
            pickle.dump(history.history, fd)
    LOGGER.info('Done!')
    return history
