"""Host input pipeline throughput: gzip HDF5 blobs (data/avc/sample.py schema) -> data_generator batches."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from l3embedding_amd import h5lite, train
n_files, per_file, batch = 3, 256, 64
d = tempfile.mkdtemp()
rng = np.random.RandomState(0)
t0 = time.time()
for i in range(n_files):
    # natural-ish content so gzip does real work (random noise would not compress at all)
    vid = (rng.randint(0, 32, size=(per_file, 224, 224, 3)) + 100).astype(np.uint8)
    aud = (rng.randn(per_file, 1, 48000) * 3000).astype(np.int16)
    lab = np.stack([rng.randint(0, 2, per_file), np.zeros(per_file, int)], 1).astype(np.int64); lab[:, 1] = 1 - lab[:, 0]
    root = h5lite.Group()
    for k, arr in (('audio', aud), ('video', vid), ('label', lab)):
        root.create_dataset(k, arr, compression='gzip')
    h5lite.write_file(os.path.join(d, 'blob%d.h5' % i), root)
print('wrote %d blobs in %.1f s, %.1f MB each' % (n_files, time.time() - t0, os.path.getsize(os.path.join(d, 'blob0.h5')) / 1e6))
for raw in (True, False):
    g = train.data_generator(d, batch_size=batch, raw=raw)
    next(g)
    t0 = time.time(); n = 0
    while n < n_files * per_file - batch:
        b = next(g); n += len(b['label'])
    dt = time.time() - t0
    print('raw=%s: %.0f pairs/s (%.1f ms per batch of %d)' % (raw, n / dt, 1e3 * dt / (n / batch), batch))
