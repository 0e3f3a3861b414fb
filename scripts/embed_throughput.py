"""Embedding export throughput (05_generate_embedding_samples.py consumer path): load_embedding().predict."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from l3embedding_amd import model
mt = 'cnn_L3_melspec2'
m, _, _ = model.MODELS[mt]()
d = tempfile.mkdtemp(); wp = os.path.join(d, 'w.h5')
m.save_weights(wp)
for B in (32, 64, 256):
    emb = model.load_embedding(wp, mt, 'audio', 'original')
    x = np.random.RandomState(0).uniform(-1, 1, size=(1024, 1, 48000)).astype(np.float32)
    emb.predict(x[:B], batch_size=B)
    t0 = time.time(); out = emb.predict(x, batch_size=B); dt = time.time() - t0
    print('audio embedding %s, engine batch %d: %.0f clips/s (%d x %d)' % (out.shape, B, len(x) / dt, len(x), out.shape[1]), flush=True)
