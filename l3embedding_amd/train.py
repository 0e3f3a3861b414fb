"""`train()` of l3embedding/train.py:218-421 on the MI355X engine.

Same call signature, defaults and run-directory artefacts as the reference, so 03_train_embedding.py
can switch imports and 05_generate_embedding_samples.py can consume a run made here:

    <output_dir>/embedding/<subset>/<model_type>/<YYYYmmddHHMMSS>/        (train.py:231-234,277)
        config.json  model_spec.pkl  model.json                           (train.py:289-313)
        model_latest.h5  model_best_valid_accuracy.h5  model_best_valid_loss.h5
        model_checkpoint.NN.h5                                             (train.py:316-355)
        history_checkpoint.pkl  history_csvlog.csv  history.pkl            (train.py:357-365,417-419)

`<subset>` is the training directory's name up to its last underscore (train.py:231-233), and
05_generate_embedding_samples.py:144-153 recovers `model_type` from that path.

The data path is `blobfeed` (shard-aware, stored dtypes to the GPU), the callbacks are `callbacks`;
the module also keeps the reference's generator entry points (`data_generator`,
`single_epoch_data_generator`, `get_restart_info`) as thin front-ends of those.

Out of scope: Google-Sheets logging, the rotating console/file logger of log.py.
"""
import datetime
import getpass
import json
import logging
import os
import pickle
import subprocess

from . import blobfeed
from .callbacks import CSVLogger, LossHistory, ModelCheckpoint, TimeHistory, last_epoch_record
from .model import MODELS, Adam, load_model

LOGGER = logging.getLogger('l3embedding')
LOGGER.setLevel(logging.DEBUG)

ARTEFACTS = dict(latest='model_latest.h5', best_acc='model_best_valid_accuracy.h5', best_loss='model_best_valid_loss.h5',
                 periodic='model_checkpoint.{epoch:02d}.h5', csv='history_csvlog.csv', loss_pkl='history_checkpoint.pkl',
                 history='history.pkl', config='config.json', spec='model_spec.pkl', json='model.json')


# ---------------------------------------------------------------------------------------------------
# reference generator entry points (train.py:142-215), served by blobfeed / callbacks
# ---------------------------------------------------------------------------------------------------
def data_generator(data_dir, batch_size=512, random_state=20180123, start_batch_idx=None, keys=None):
    """Endless `{'audio','video','label'}` batches in the reference's batch sequence (see `blobfeed`).
    Rows keep their stored dtypes; the [-1,1] scalings of train.py:186,189 are applied by the engine."""
    return blobfeed.BlobFeed(data_dir, batch_size, random_state, start_batch_idx, keys)


def single_epoch_data_generator(data_dir, epoch_size, **kwargs):
    return blobfeed.RestartingFeed(lambda: data_generator(data_dir, **kwargs), epoch_size)


def get_restart_info(history_path):
    return last_epoch_record(history_path)


# ---------------------------------------------------------------------------------------------------
# run directory
# ---------------------------------------------------------------------------------------------------
def model_id_for(train_data_dir, model_type):
    """`<subset>/<model_type>`: subset = basename of the training directory cut at its last '_'
    (e.g. `music_train` -> `music`), train.py:231-234.  Like the reference, a name without '_' is an error."""
    subset = os.path.basename(train_data_dir)
    cut = subset.rfind('_')
    if cut < 0:
        raise ValueError('substring not found: train_data_dir basename "{}" has no "_"'.format(subset))
    return os.path.join(subset[:cut], model_type)


def embedding_desc_str(model_dir):
    """How 05_generate_embedding_samples.py:144-153 names a run: the path below `embedding/` without the
    timestamp; its last component must be the model type."""
    parts = os.path.normpath(model_dir).split(os.sep)
    return '/'.join(parts[parts.index('embedding') + 1:-1])


def _git_commit():
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        return subprocess.check_output(['git', 'rev-parse', 'HEAD'], cwd=here, stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def _world():
    """(rank, world size, is-distributed) of this process; (0, 1, False) without torch.distributed."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), True
    except ImportError:
        pass
    return 0, 1, False


def _agree_on(value):
    """Rank 0's value on every rank (the timestamped directory name must not differ between ranks)."""
    import torch.distributed as dist
    box = [value]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def _checkpoint_callbacks(model_dir, checkpoint_interval, resume):
    """The four weight-file writers of train.py:316-355; `resume` = (last epoch, val_acc, val_loss) or None."""
    path = lambda key: os.path.join(model_dir, ARTEFACTS[key])
    latest = ModelCheckpoint(path('latest'), verbose=1, logger=LOGGER)
    best_acc = ModelCheckpoint(path('best_acc'), monitor='val_acc', save_best_only=True, verbose=1, logger=LOGGER)
    best_loss = ModelCheckpoint(path('best_loss'), monitor='val_loss', save_best_only=True, verbose=1, logger=LOGGER)
    periodic = ModelCheckpoint(path('periodic'), period=checkpoint_interval, logger=LOGGER)
    if resume is not None:
        last_epoch, best_acc.best, best_loss.best = resume
        periodic.epochs_since_last_save = (last_epoch + 1) % checkpoint_interval
    return [latest, best_acc, best_loss, periodic]


def train(train_data_dir, validation_data_dir, output_dir,
          num_epochs=150, train_epoch_size=512, validation_epoch_size=1024,
          train_batch_size=64, validation_batch_size=64,
          model_type='cnn_L3_orig', random_state=20180123,
          learning_rate=1e-4, verbose=False, checkpoint_interval=10,
          log_path=None, disable_logging=False, gpus=1, continue_model_dir=None,
          gsheet_id=None, google_dev_app_name=None):
    """Train an AVC model (train.py:218-421).  With `gpus > 1` this is one rank of a
    `torch.distributed` job (one process per GPU): every rank trains on its shard of each batch, rank 0
    owns the run directory."""
    call_args = dict(locals())
    if not disable_logging and log_path:
        handler = logging.FileHandler(log_path)
        handler.setFormatter(logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s'))
        LOGGER.addHandler(handler)
    if gsheet_id:
        LOGGER.warning('Google-Sheets logging is out of scope for this build; ignoring gsheet_id')

    model_id = model_id_for(train_data_dir, model_type)
    resume = None
    if continue_model_dir:
        model_dir = continue_model_dir
        resume = get_restart_info(os.path.join(model_dir, ARTEFACTS['csv']))
        m, _, _ = load_model(os.path.join(model_dir, ARTEFACTS['latest']), model_type, return_io=True, src_num_gpus=gpus)
    else:
        model_dir = os.path.join(output_dir, 'embedding', model_id, datetime.datetime.now().strftime('%Y%m%d%H%M%S'))
        m, _, _ = MODELS[model_type](num_gpus=gpus)

    rank, world, distributed = _world()
    if distributed:
        model_dir = _agree_on(model_dir)
    writes = rank == 0
    if writes:
        os.makedirs(model_dir, exist_ok=True)

    # NOTE from the reference: this loss is summed over both one-hot columns (train.py:269)
    m.compile(Adam(lr=learning_rate), loss='categorical_crossentropy', metrics=['accuracy'])
    LOGGER.info('Model files can be found in "%s"', model_dir)

    config = dict(call_args, username=getpass.getuser(), model_id=model_id, model_dir=model_dir, git_commit=_git_commit(),
                  backend='libl3hip (MI355X)')
    LOGGER.info('Training with the following arguments: %s', config)
    if writes:
        with open(os.path.join(model_dir, ARTEFACTS['config']), 'w') as fh:
            json.dump(config, fh, indent=2)
        with open(os.path.join(model_dir, ARTEFACTS['spec']), 'wb') as fh:
            pickle.dump(m.get_config(), fh)
        with open(os.path.join(model_dir, ARTEFACTS['json']), 'w') as fh:
            json.dump(m.to_json(), fh, indent=2)

    timer = TimeHistory(logger=LOGGER)
    callbacks = [timer]
    if writes:      # every rank holds identical weights and logs: one writer
        callbacks = _checkpoint_callbacks(model_dir, checkpoint_interval, resume) + [
            timer, LossHistory(os.path.join(model_dir, ARTEFACTS['loss_pkl'])),
            CSVLogger(os.path.join(model_dir, ARTEFACTS['csv']), append=True, separator=',')]

    first_epoch = resume[0] + 1 if resume is not None else 0
    shard = dict(rank=rank, world=world) if gpus > 1 and distributed else {}
    LOGGER.info('Setting up train data generator...')
    train_feed = blobfeed.BlobFeed(train_data_dir, train_batch_size, random_state,
                                   start_batch_idx=train_epoch_size * first_epoch if resume is not None else None, **shard)
    LOGGER.info('Setting up validation data generator...')
    val_feed = blobfeed.RestartingFeed(
        lambda: blobfeed.BlobFeed(validation_data_dir, validation_batch_size, random_state, **shard), validation_epoch_size)

    LOGGER.info('Fitting model...')
    try:
        history = m.fit_generator(blobfeed.as_model_inputs(train_feed, train_batch_size, sharded=bool(shard)),
                                  train_epoch_size, num_epochs,
                                  validation_data=blobfeed.as_model_inputs(val_feed, validation_batch_size, sharded=bool(shard)),
                                  validation_steps=validation_epoch_size, callbacks=callbacks,
                                  verbose=1 if verbose else 2, initial_epoch=first_epoch)
    finally:
        train_feed.close()          # the feeds keep their last few blobs open (memory-mapped): released with the run
        val_feed.close()

    LOGGER.info('Done training. Saving results to disk...')
    if writes:
        with open(os.path.join(model_dir, ARTEFACTS['history']), 'wb') as fh:
            pickle.dump(history.history, fh)
    LOGGER.info('Done!')
    return history
