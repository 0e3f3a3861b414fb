"""End-to-end real-data path: gzip HDF5 blobs -> blobfeed.BlobFeed -> prefetch thread -> fit_generator; the per-rank
decode rate of a sharded feed (rank 0 of 1, 2, 4, 8, alone on the host); and -- `concurrent` argument -- all eight
ranks' readers of a world-8 job at the same time, one process each (what an 8-GPU node's host actually has to sustain)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from l3embedding_amd import blobfeed, h5lite, model
n_files, per_file, batch, steps = 8, 256, 64, 150
if sys.argv[1:2] == ['reader']:                     # child of the `concurrent` mode: one rank's reader on existing blobs
    d_, rank, world = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    g = blobfeed.BlobFeed(d_, batch * world, rank=rank, world=world)
    next(g)
    t0 = time.time()
    n = 0
    for _ in range(20):
        n += len(next(g)['label'])
    print('READER %d %.4f' % (n, time.time() - t0))
    sys.exit(0)
d = tempfile.mkdtemp()
rng = np.random.RandomState(0)
for i in range(n_files):
    vid = (rng.randint(0, 32, size=(per_file, 224, 224, 3)) + 100).astype(np.uint8)
    aud = (rng.randn(per_file, 1, 48000) * 3000).astype(np.int16)
    lab0 = rng.randint(0, 2, per_file)
    lab = np.stack([lab0, 1 - lab0], 1).astype(np.int64)
    root = h5lite.Group()
    for k, arr in (('audio', aud), ('video', vid), ('label', lab)):
        root.create_dataset(k, arr, compression='gzip')
    h5lite.write_file(os.path.join(d, 'blob%d.h5' % i), root)
print('blobs written', flush=True)
for world in (1, 2, 4, 8):
    g = blobfeed.BlobFeed(d, batch * world, rank=0, world=world)
    next(g); t0 = time.time(); n = 0
    for _ in range(20):
        n += len(next(g)['label'])
    dt = time.time() - t0
    print('feed alone, rank 0 of %d (global batch %d): %.0f local pairs/s = %.0f global pairs/s' %
          (world, batch * world, n / dt, n * world / dt), flush=True)

if 'concurrent' in sys.argv[1:]:
    # eight separate interpreter processes started together (a fork of this one would inherit h5lite's reader threads)
    import subprocess
    world = 8
    d_raw = tempfile.mkdtemp()
    blobfeed.rewrite_uncompressed(d, d_raw)          # the feed option for CPU-starved hosts: same blobs, contiguous datasets
    for tag, dd in (('gzip blobs (as the reference writes them)', d), ('uncompressed blobs (blobfeed.rewrite_uncompressed)', d_raw)):
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), 'reader', dd, str(r), str(world)],
                                  stdout=subprocess.PIPE) for r in range(world)]
        outs = [p.communicate(timeout=500)[0].decode().strip().splitlines()[-1].split() for p in procs]
        wall = time.time() - t0
        res = [(int(o[1]), float(o[2])) for o in outs]
        print('8 readers at the same time, %s (ranks 0..7 of 8, global batch %d, one process each): per rank %s pairs/s; aggregate '
              '%.0f pairs/s while all eight run (%.1f s wall incl. interpreter start and each rank\'s first batch)' %
              (tag, batch * world, ' '.join('%.0f' % (n / dt) for n, dt in res), sum(n for n, _ in res) / max(dt for _, dt in res), wall),
              flush=True)
m, inputs, outputs = model.MODELS['cnn_L3_melspec2']()
m.compile(model.Adam(lr=1e-4), loss='categorical_crossentropy', metrics=['accuracy'])
dirs = [('gzip blobs', d)]
if 'concurrent' in sys.argv[1:]:
    dirs.append(('uncompressed blobs', d_raw))
for tag, dd in dirs:
    for depth in (10, 0):
        gen = blobfeed.as_model_inputs(blobfeed.BlobFeed(dd, batch))
        m.fit_generator(gen, 3, 1, verbose=0, max_queue_size=depth)          # warm-up
        t0 = time.time()
        m.fit_generator(gen, steps, 1, verbose=0, max_queue_size=depth)
        dt = time.time() - t0
        print('fit_generator on %s (prefetch depth %d): %.0f pairs/s, %.1f ms/step' % (tag, depth, steps * batch / dt, 1e3 * dt / steps), flush=True)
