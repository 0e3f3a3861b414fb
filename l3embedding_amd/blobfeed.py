"""Batch feed over the AVC HDF5 batch blobs, shard-aware.

Replaces the reference's pescador/h5py generator (l3embedding/train.py:134-205).  What is kept is
its observable *batch sequence*:

  * file order: one pass over `os.listdir(data_dir)` as listed, then the list is reshuffled before
    every further pass by a Mersenne-Twister stream seeded with `random_state` (train.py:134-139,146,154);
  * a batch is `batch_size` consecutive rows of that concatenated stream, so batches spill over blob
    boundaries and a blob's tail joins the next blob's head (train.py:161-176); rows per blob =
    `len(blob['label'])` (train.py:159);
  * `start_batch_idx` drops that many leading batches (the resume skip, train.py:164-165,181-193);
  * only the keys `audio`, `video`, `label` are delivered (train.py:149-151);
  * the validation feed restarts from the top after `epoch_size` batches (train.py:198-205).

What is different by design:

  * two passes.  `plan_batches` turns the file order into per-batch lists of `(path, lo, hi)` row
    spans from the blob *headers* alone; data is touched only for the spans a consumer asks for.  The
    resume skip therefore inflates nothing;
  * rank r of a data-parallel job reads -- and inflates, chunk by chunk (`h5lite.Dataset.read_rows`) --
    only rows `get_slice_bounds(batch, world, r)` of every batch (training_utils.py:121-133 arithmetic),
    so decode cost per rank falls as 1/world instead of every rank inflating the global batch;
  * rows are delivered in their stored dtypes (uint8 frames, int16 PCM, int labels): the scalings of
    train.py:186,189 run on the GPU (`l3_upload_batch_raw`, bit-exact), which cuts host->device bytes 3.2x;
  * blobs may be stored UNCOMPRESSED (`rewrite_uncompressed`, `python -m l3embedding_amd.blobfeed uncompress SRC DST`): the reference
    writes them gzip-ed (data/avc/sample.py:565-568), and inflating is what bounds the feed -- eight ranks' readers deliver 15-20 k
    pairs/s in total under a 16-core CPU quota, half of what eight mixed-precision engines consume.  A contiguous
    dataset is a slice of the memory-mapped file: same batches (the feed never looks at the storage layout), 3.2x the bytes on disk;
  * the shuffle stream is private to the feed (`random.Random(random_state)`), not the process-global
    `random` module the reference reseeds from two generators at once; the first reshuffle is the one
    `random.seed(random_state); random.shuffle(lst)` produces.
"""
import os
import random
from collections import OrderedDict, namedtuple

import numpy as np

from . import h5lite
from .training_utils import get_slice_bounds

FEED_KEYS = ('audio', 'video', 'label')
Span = namedtuple('Span', 'path lo hi')          # rows [lo, hi) of one blob


def blob_order(data_dir, random_state=20180123, shuffle=True):
    """Endless sequence of blob paths: listing order first, reshuffled before each later pass."""
    names = list(os.listdir(data_dir))
    if not names:
        raise ValueError('no batch blobs in "{}"'.format(data_dir))
    rng = random.Random(random_state)
    while True:
        for name in names:
            yield os.path.join(data_dir, name)
        if shuffle:
            rng.shuffle(names)


def plan_batches(paths, batch_size, rows_of):
    """Cuts the row stream of `paths` into batches: yields a list of Spans per batch.
    `rows_of(path)` gives a blob's row count (header read only)."""
    if batch_size < 1:
        raise ValueError('batch_size must be positive')
    spans, missing = [], batch_size
    for path in paths:
        total, at = rows_of(path), 0
        while at < total:
            take = min(missing, total - at)
            spans.append(Span(path, at, at + take))
            at += take
            missing -= take
            if missing == 0:
                yield spans
                spans, missing = [], batch_size


class BlobReader(object):
    """Keeps the last few blobs open (consecutive batches come from the same file) and counts what it reads."""

    def __init__(self, max_open=3):
        self._open = OrderedDict()
        self._max_open = max_open
        self._rows = {}
        self.rows_read = 0           # rows actually fetched (per key 'label'), for the 1/world decode-cost test

    def _file(self, path):
        f = self._open.pop(path, None)
        if f is None:
            f = h5lite.File(path)
            while len(self._open) >= self._max_open:
                self._open.popitem(last=False)[1].close()
        self._open[path] = f
        return f

    def rows_of(self, path):
        n = self._rows.get(path)
        if n is None:
            n = self._rows[path] = len(self._file(path)['label'])
        return n

    def read(self, span, keys):
        f = self._file(span.path)
        self.rows_read += span.hi - span.lo
        return {k: f[k].read_rows(span.lo, span.hi) for k in keys}

    def close(self):
        while self._open:
            self._open.popitem()[1].close()


def _clip(spans, lo, hi):
    """The parts of a batch's spans that fall into batch rows [lo, hi)."""
    out, at = [], 0
    for sp in spans:
        n = sp.hi - sp.lo
        a, b = max(lo, at), min(hi, at + n)
        if a < b:
            out.append(Span(sp.path, sp.lo + (a - at), sp.lo + (b - at)))
        at += n
    return out


class BlobFeed(object):
    """Iterator of batches `{'audio', 'video', 'label'}` (stored dtypes) for one rank.

    rank / world: this consumer's shard; every batch then holds rows
    `get_slice_bounds(batch_size, world, rank)` of the global batch and `feed.global_batch == batch_size`.
    """

    def __init__(self, data_dir, batch_size=512, random_state=20180123, start_batch_idx=None, keys=None,
                 rank=0, world=1, shuffle=True):
        if not 0 <= rank < world:
            raise ValueError('rank %d outside world %d' % (rank, world))
        self.keys = tuple(keys) if keys else FEED_KEYS
        self.global_batch = int(batch_size)
        self.bounds = get_slice_bounds(self.global_batch, world, rank)
        if self.bounds[1] <= self.bounds[0]:
            raise ValueError('batch_size %d leaves rank %d of %d without rows (training_utils.py:121-133: '
                             'step = batch_size // gpus); use a batch of at least %d'
                             % (self.global_batch, rank, world, world))
        self.reader = BlobReader()
        self._plan = plan_batches(blob_order(data_dir, random_state, shuffle), self.global_batch, self.reader.rows_of)
        self.batch_idx = 0
        for _ in range(int(start_batch_idx or 0)):      # resume: headers only, no rows are read
            next(self._plan)
            self.batch_idx += 1

    def __iter__(self):
        return self

    def __next__(self):
        spans = _clip(next(self._plan), *self.bounds)
        self.batch_idx += 1
        parts = [self.reader.read(sp, self.keys) for sp in spans]
        if len(parts) == 1:
            return parts[0]
        return {k: np.concatenate([p[k] for p in parts]) for k in self.keys}

    def close(self):
        self.reader.close()


class RestartingFeed(object):
    """`epoch_size` batches from a fresh feed, then a new feed from the top, forever -- the validation
    generator of train.py:198-205 (every validation pass sees the same batches)."""

    def __init__(self, make_feed, epoch_size):
        if epoch_size < 1:
            raise ValueError('epoch_size must be positive')
        self._make, self._epoch_size = make_feed, int(epoch_size)
        self._feed, self._left = None, 0

    def __iter__(self):
        return self

    def __next__(self):
        if self._left == 0:
            if self._feed is not None:
                self._feed.close()
            self._feed, self._left = self._make(), self._epoch_size
        self._left -= 1
        return next(self._feed)

    def close(self):
        if self._feed is not None:
            self._feed.close()
            self._feed, self._left = None, 0


class ShardedInputs(list):
    """`[video, audio]` of ONE rank's shard; `global_batch` tells the model the batch is already split
    (so it must not slice it again) and how large the concatenated batch is (loss scaling)."""

    def __init__(self, arrays, global_batch):
        super().__init__(arrays)
        self.global_batch = int(global_batch)


def as_model_inputs(feed, global_batch=None, sharded=False):
    """Batch dicts -> the `([video, audio], label)` tuples fit_generator consumes (feed order of
    train.py:382-384: inputs ['video', 'audio'], target 'label')."""
    try:
        for batch in feed:
            x = [batch['video'], batch['audio']]
            yield (ShardedInputs(x, global_batch) if sharded else x), batch['label']
    finally:
        closer = getattr(feed, 'close', None)           # generator closed (end of fit_generator): release the open blobs
        if closer is not None:
            closer()


def rewrite_uncompressed(src_dir, dst_dir, keys=None):
    """Every blob of `src_dir` rewritten into `dst_dir` with contiguous (unfiltered) datasets -- same names, shapes, dtypes and
    values, so `BlobFeed(dst_dir, ...)` delivers exactly the batches of `BlobFeed(src_dir, ...)` (the file order is os.listdir's,
    so keep the names), without an inflate on the way.  `keys`: datasets to keep (default: all top-level ones)."""
    os.makedirs(dst_dir, exist_ok=True)
    done = []
    for name in os.listdir(src_dir):
        path = os.path.join(src_dir, name)
        if not os.path.isfile(path):
            continue
        with h5lite.File(path) as f:
            root = h5lite.Group()
            for k, node in f.root.children.items():
                if keys is not None and k not in keys:
                    continue
                if isinstance(node, h5lite.Dataset):
                    root.create_dataset(k, node.read())
            for k, v in f.root.attrs.items():
                root.attrs[k] = v
        tmp = os.path.join(dst_dir, name + '.partial.%d' % os.getpid())
        h5lite.write_file(tmp, root)
        os.replace(tmp, os.path.join(dst_dir, name))
        done.append(name)
    return done


if __name__ == '__main__':
    import sys
    if len(sys.argv) == 4 and sys.argv[1] == 'uncompress':
        print('%d blobs rewritten' % len(rewrite_uncompressed(sys.argv[2], sys.argv[3])))
    else:
        raise SystemExit('usage: python -m l3embedding_amd.blobfeed uncompress SRC_DIR DST_DIR')
