"""CPU tests of the host side of the training entry point: the shard-aware blob feed (batch sequence of
l3embedding/train.py:134-205 pinned by known answers worked out by hand below), the callbacks' artefacts
(train.py:316-365) and the run-directory bookkeeping (train.py:231-234,277 and its consumer
05_generate_embedding_samples.py:144-153).  The reference module cannot be imported here (keras,
pescador, h5py absent: SURVEY.md 8(c)), so nothing below runs or shares its code."""
import os
import pickle
import random

import numpy as np
import pytest

from l3embedding_amd import blobfeed, callbacks, h5lite, train as T
from l3embedding_amd.training_utils import get_slice_bounds

A_T, V_HW = 48, 4          # tiny stand-ins for 48000 samples / 224x224 frames: the feed is shape-agnostic


def _write_blob(path, file_id, rows, extra_keys=True):
    """Row r of file f carries label [1000*f + r, -(1000*f + r)] and audio/video filled with the same id,
    so a delivered batch names exactly which rows it holds."""
    ids = 1000 * file_id + np.arange(rows)
    root = h5lite.Group()
    root.create_dataset('audio', np.broadcast_to(ids[:, None, None], (rows, 1, A_T)).astype(np.int16), compression='gzip')
    root.create_dataset('video', np.broadcast_to((ids % 251)[:, None, None, None], (rows, V_HW, V_HW, 3)).astype(np.uint8),
                        compression='gzip')
    root.create_dataset('label', np.stack([ids, -ids], 1).astype(np.int64), compression='gzip')
    if extra_keys:          # metadata fields of data/avc/sample.py:371-386 that must not reach the batch
        root.create_dataset('audio_start_sample_idx', np.arange(rows))
    h5lite.write_file(path, root)


@pytest.fixture
def blob_dir(tmp_path, monkeypatch):
    """Three blobs of 5, 3 and 7 rows with a pinned directory listing order (b, a, c)."""
    d = str(tmp_path / 'subset_train')
    os.makedirs(d)
    sizes = {'b.h5': (2, 5), 'a.h5': (1, 3), 'c.h5': (3, 7)}
    for name, (fid, rows) in sizes.items():
        _write_blob(os.path.join(d, name), fid, rows)
    real = os.listdir
    monkeypatch.setattr(os, 'listdir', lambda p: ['b.h5', 'a.h5', 'c.h5'] if os.path.abspath(p) == os.path.abspath(d) else real(p))
    return d


def _ids(batch):
    return batch['label'][:, 0].tolist()


def test_batch_sequence_known_answers(blob_dir):
    """batch_size 4 over blobs of 5 | 3 | 7 rows in listing order b, a, c:
        batch 0 = b0..b3          batch 1 = b4, a0..a2  (tail of one blob joins the next)
        batch 2 = c0..c3          batch 3 = c4..c6 + first row of the next pass
    and the second pass runs over the list reshuffled by MT19937(random_state)."""
    feed = T.data_generator(blob_dir, batch_size=4, random_state=7)
    got = [next(feed) for _ in range(4)]
    assert set(got[0]) == {'audio', 'video', 'label'}                          # train.py:149-151
    assert _ids(got[0]) == [2000, 2001, 2002, 2003]
    assert _ids(got[1]) == [2004, 1000, 1001, 1002]
    assert _ids(got[2]) == [3000, 3001, 3002, 3003]
    second = ['b.h5', 'a.h5', 'c.h5']
    random.Random(7).shuffle(second)                                           # pass 2 order
    first_of = {'b.h5': 2000, 'a.h5': 1000, 'c.h5': 3000}
    assert _ids(got[3]) == [3004, 3005, 3006, first_of[second[0]]]
    # stored dtypes reach the consumer (the GPU applies train.py:186,189), rows stay aligned across keys
    assert got[1]['audio'].dtype == np.int16 and got[1]['video'].dtype == np.uint8 and got[1]['label'].dtype == np.int64
    assert got[1]['audio'].shape == (4, 1, A_T) and got[1]['video'].shape == (4, V_HW, V_HW, 3)
    assert got[1]['audio'][:, 0, 0].tolist() == [2004, 1000, 1001, 1002]
    assert got[1]['video'][:, 0, 0, 0].tolist() == [2004 % 251, 1000 % 251, 1001 % 251, 1002 % 251]
    # the private shuffle stream is the one `random.seed(rs); random.shuffle(lst)` (train.py:146,138) produces
    ref = ['b.h5', 'a.h5', 'c.h5']
    random.seed(7)
    random.shuffle(ref)
    assert ref == second


def test_reshuffled_passes_visit_every_row_once(blob_dir):
    feed = T.data_generator(blob_dir, batch_size=5, random_state=3)
    rows = [i for _ in range(9) for i in _ids(next(feed))]                      # 45 rows = three passes of 15
    universe = sorted([2000 + i for i in range(5)] + [1000 + i for i in range(3)] + [3000 + i for i in range(7)])
    for p in range(3):
        assert sorted(rows[15 * p:15 * (p + 1)]) == universe
    assert rows[:15] == [2000, 2001, 2002, 2003, 2004, 1000, 1001, 1002, 3000, 3001, 3002, 3003, 3004, 3005, 3006]


def test_resume_skip_reads_no_rows(blob_dir):
    """start_batch_idx = N delivers batch N of the uninterrupted sequence (train.py:164-193) -- and, unlike the
    reference, inflates nothing on the way there."""
    full = T.data_generator(blob_dir, batch_size=4, random_state=11)
    want = [_ids(next(full)) for _ in range(7)]
    for skip in (1, 3, 6):
        feed = T.data_generator(blob_dir, batch_size=4, random_state=11, start_batch_idx=skip)
        assert feed.reader.rows_read == 0
        assert _ids(next(feed)) == want[skip]
        assert feed.reader.rows_read == 4
    assert _ids(next(T.data_generator(blob_dir, batch_size=4, random_state=11, start_batch_idx=None))) == want[0]


def test_validation_feed_restarts_each_epoch(blob_dir):
    se = T.single_epoch_data_generator(blob_dir, 2, batch_size=5, random_state=1)
    e = [_ids(next(se)) for _ in range(6)]
    assert e[0] == e[2] == e[4] and e[1] == e[3] == e[5] and e[0] != e[1]      # train.py:198-205
    with pytest.raises(ValueError):
        T.single_epoch_data_generator(blob_dir, 0)


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_feeds_tile_the_global_batch_and_split_the_decode_cost(blob_dir, world):
    """Rank r gets rows get_slice_bounds(batch, world, r) of every batch (training_utils.py:121-133: the last
    rank takes the remainder) and reads only those rows: summed over ranks = one global read."""
    B = 7
    whole = blobfeed.BlobFeed(blob_dir, B, 5)
    shards = [blobfeed.BlobFeed(blob_dir, B, 5, rank=r, world=world) for r in range(world)]
    for _ in range(6):
        want = next(whole)
        parts = [next(s) for s in shards]
        for r, p in enumerate(parts):
            lo, hi = get_slice_bounds(B, world, r)
            assert len(p['label']) == hi - lo
            for k in want:
                assert np.array_equal(p[k], want[k][lo:hi]), (r, k)
        assert np.array_equal(np.concatenate([p['audio'] for p in parts]), want['audio'])
    assert sum(s.reader.rows_read for s in shards) == whole.reader.rows_read == 6 * B
    assert max(s.reader.rows_read for s in shards) <= 6 * (B - (world - 1) * (B // world))
    with pytest.raises(ValueError):
        blobfeed.BlobFeed(blob_dir, B, rank=2, world=2)
    x, y = next(blobfeed.as_model_inputs(shards[0], B, sharded=True))
    assert x.global_batch == B and len(x) == 2 and x[0].dtype == np.uint8 and x[1].dtype == np.int16   # [video, audio]


def test_feed_edge_cases(tmp_path):
    empty = tmp_path / 'none_train'
    empty.mkdir()
    with pytest.raises(ValueError, match='no batch blobs'):
        next(T.data_generator(str(empty), batch_size=2))
    d = tmp_path / 'x_train'
    d.mkdir()
    _write_blob(str(d / 'only.h5'), 4, 3)
    _write_blob(str(d / 'zero.h5'), 5, 0)                       # an empty blob contributes nothing
    feed = T.data_generator(str(d), batch_size=7, random_state=0)           # larger than one pass: wraps around
    assert _ids(next(feed)) == [4000, 4001, 4002, 4000, 4001, 4002, 4000]
    assert _ids(next(feed)) == [4001, 4002, 4000, 4001, 4002, 4000, 4001]
    with pytest.raises(ValueError):
        next(blobfeed.plan_batches(iter([]), 0, lambda p: 1))


def test_chunked_partial_reads_inflate_only_what_is_needed(tmp_path):
    """h5lite.Dataset.read_rows on a gzip dataset with several chunks along the batch axis."""
    rng = np.random.RandomState(0)
    arr = rng.randint(0, 256, (40, 64, 64, 3)).astype(np.uint8)          # 12 KiB per row
    root = h5lite.Group()
    root.create_dataset('video', arr, compression='gzip')
    root.create_dataset('label', np.arange(80).reshape(40, 2))
    p = str(tmp_path / 'v.h5')
    h5lite.write_file(p, root)
    with h5lite.File(p) as f:
        ds = f['video']
        assert ds.shape == arr.shape and ds.dtype == np.uint8 and len(f['label']) == 40 and 'video' in f and 'nope' not in f
        for lo, hi in [(0, 40), (3, 4), (17, 29), (39, 40), (5, 5), (30, 99)]:
            assert np.array_equal(ds.read_rows(lo, hi), arr[lo:hi]), (lo, hi)
        assert np.array_equal(f['label'].read_rows(10, 12), np.arange(80).reshape(40, 2)[10:12])
    assert np.array_equal(h5lite.read_file(p)['video'], arr)


# ---- callbacks ---------------------------------------------------------------------------------------------
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
    th = T.TimeHistory()
    cbs = [latest, best_acc, best_loss, every, csvl, lh, th]
    for c in cbs:
        c.set_model(m)
        c.on_train_begin({})
    logs = [dict(loss=1.0, acc=0.5, val_loss=0.9, val_acc=0.55), dict(loss=0.8, acc=0.6, val_loss=1.1, val_acc=0.60),
            dict(loss=0.7, acc=0.7, val_loss=0.8, val_acc=0.58)]
    for ep, lg in enumerate(logs):
        for c in cbs:
            c.on_epoch_begin(ep, {})
            c.on_batch_begin(0, {})
            c.on_batch_end(0, {})
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
    with open(str(tmp_path / 'history_checkpoint.pkl'), 'rb') as fh:          # train.py:50-53
        assert pickle.load(fh) == {'loss': [1.0, 0.8, 0.7], 'val_loss': [0.9, 1.1, 0.8]}
    assert lh.loss == [1.0, 0.8, 0.7] and len(th.epoch_times) == 3 and len(th.batch_times) == 3
    # resuming appends without a second header (train.py:363-365 append=True)
    csv2 = T.CSVLogger(str(tmp_path / 'history_csvlog.csv'), append=True)
    csv2.set_model(m)
    csv2.on_train_begin({})
    csv2.on_epoch_end(3, logs[0])
    csv2.on_train_end({})
    rows = open(str(tmp_path / 'history_csvlog.csv')).read().strip().split('\n')
    assert len(rows) == 5 and rows.count('epoch,acc,loss,val_acc,val_loss') == 1
    with pytest.raises(ValueError):
        (tmp_path / 'empty.csv').write_text('epoch,acc\n')
        callbacks.last_epoch_record(str(tmp_path / 'empty.csv'))


def test_resumed_checkpoint_callbacks_are_seeded(tmp_path):
    """train.py:333-334,342-343,352-353: a continued run keeps the best-so-far thresholds and the periodic phase."""
    cbs = T._checkpoint_callbacks(str(tmp_path), 10, resume=(13, 0.71, 0.52))
    latest, best_acc, best_loss, periodic = cbs
    assert best_acc.best == 0.71 and best_loss.best == 0.52 and periodic.epochs_since_last_save == 4
    m = _FakeModel()
    for c in cbs:
        c.set_model(m)
    for c in cbs:
        c.on_epoch_end(14, dict(val_acc=0.70, val_loss=0.50))
    assert m.saved == ['model_latest.h5', 'model_best_valid_loss.h5']              # acc did not improve, loss did
    fresh = T._checkpoint_callbacks(str(tmp_path), 10, resume=None)
    assert fresh[1].best == -np.inf and fresh[2].best == np.inf and fresh[3].epochs_since_last_save == 0


# ---- run directory --------------------------------------------------------------------------------------------
def test_run_directory_layout_is_what_05_generate_embedding_samples_parses():
    """train.py:231-234: model_id = <basename up to the last '_'>/<model_type>; 05_generate_embedding_samples.py
    :144-153 strips everything up to 'embedding/' and the timestamp and reads model_type from the end."""
    assert T.model_id_for('/scratch/data/music_train', 'cnn_L3_melspec2') == os.path.join('music', 'cnn_L3_melspec2')
    assert T.model_id_for('/data/audioset_music_v2_train', 'cnn_L3_orig') == os.path.join('audioset_music_v2', 'cnn_L3_orig')
    with pytest.raises(ValueError, match='substring not found'):
        T.model_id_for('/data/train', 'cnn_L3_orig')                 # str.rindex raises the same in the reference
    model_dir = os.path.join('/out', 'embedding', T.model_id_for('/d/music_train', 'cnn_L3_melspec2'), '20180223100000')
    # the consumer's own parsing, restated: model_desc_start_idx = path.index('/embedding/') + len('/embedding/');
    # model_desc_end_idx = path.rindex('/'); embedding_desc_str = path[start:end]; model_type = desc.split('/')[-1]
    weights = os.path.join(model_dir, 'model_best_valid_accuracy.h5')
    start = weights.index('/embedding/') + len('/embedding/')
    desc = os.path.dirname(weights)[start:]
    desc = desc[:desc.rindex('/')]
    assert desc == 'music/cnn_L3_melspec2' == T.embedding_desc_str(model_dir)
    assert desc.split('/')[-1] == 'cnn_L3_melspec2'


def test_train_defaults_match_reference_signature():
    import inspect
    sig = inspect.signature(T.train)
    d = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect.Parameter.empty}
    assert list(sig.parameters)[:3] == ['train_data_dir', 'validation_data_dir', 'output_dir']
    assert d == dict(num_epochs=150, train_epoch_size=512, validation_epoch_size=1024, train_batch_size=64,
                     validation_batch_size=64, model_type='cnn_L3_orig', random_state=20180123, learning_rate=1e-4,
                     verbose=False, checkpoint_interval=10, log_path=None, disable_logging=False, gpus=1,
                     continue_model_dir=None, gsheet_id=None, google_dev_app_name=None)        # train.py:218-225


def test_feed_errors_and_cleanup(tmp_path):
    """A rank without rows is a clear error (not an opaque concatenate of nothing), and closing the generator
    fit_generator consumed closes the feed's open blobs."""
    from l3embedding_amd import blobfeed
    d = tmp_path / 'blobs'
    d.mkdir()
    _write_blob(str(d / 'a.h5'), 0, 6)
    with pytest.raises(ValueError, match='leaves rank 0 of 4 without rows'):
        blobfeed.BlobFeed(str(d), batch_size=3, rank=0, world=4)
    feed = blobfeed.BlobFeed(str(d), batch_size=2)
    gen = blobfeed.as_model_inputs(feed)
    next(gen)
    assert len(feed.reader._open) == 1
    gen.close()
    assert len(feed.reader._open) == 0
    rf = blobfeed.RestartingFeed(lambda: blobfeed.BlobFeed(str(d), batch_size=2), 2)
    next(rf)
    inner = rf._feed
    rf.close()
    assert len(inner.reader._open) == 0


def test_uncompressed_blobs_deliver_the_same_batches(blob_dir, tmp_path, monkeypatch):
    """blobfeed.rewrite_uncompressed (the feed option for hosts whose CPU quota cannot inflate fast enough): same names,
    same datasets stored contiguously -- the feed delivers bit-identical batches over three reshuffled passes, sharded or not, and
    the reference-written gzip blobs of tests/golden/ref_blobs survive the round trip too."""
    dst = str(tmp_path / 'subset_train_raw')
    names = blobfeed.rewrite_uncompressed(blob_dir, dst)
    assert sorted(names) == ['a.h5', 'b.h5', 'c.h5']
    real = os.listdir
    monkeypatch.setattr(os, 'listdir', lambda p: ['b.h5', 'a.h5', 'c.h5'] if os.path.abspath(p) in (os.path.abspath(dst), os.path.abspath(blob_dir)) else real(p))
    with h5lite.File(os.path.join(dst, 'b.h5')) as f:
        assert f['video']._cls == 1 and not f['video']._filters          # contiguous layout, no filter pipeline
    for rank, world in ((0, 1), (1, 2)):
        ga = blobfeed.BlobFeed(blob_dir, 4, random_state=3, rank=rank, world=world)
        gb = blobfeed.BlobFeed(dst, 4, random_state=3, rank=rank, world=world)
        for _ in range(12):
            a, b = next(ga), next(gb)
            assert sorted(a) == sorted(b) == ['audio', 'label', 'video']
            assert all(a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]) for k in a)
        ga.close()
        gb.close()
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_blobs')
    if os.path.isdir(ref) and os.listdir(ref):
        dst2 = str(tmp_path / 'ref_raw')
        for name in blobfeed.rewrite_uncompressed(ref, dst2):
            with h5lite.File(os.path.join(ref, name)) as fa, h5lite.File(os.path.join(dst2, name)) as fb:
                for k in ('audio', 'video', 'label'):
                    assert np.array_equal(fa[k].read(), fb[k].read())
