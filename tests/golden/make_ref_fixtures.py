"""Fixtures computed by the REFERENCE's own code (marl/l3embedding under /root/reference), not by this repo's oracle.

Run in the build container only (the GPU box has no /root/reference):

    /opt/conda/bin/python3.9 tests/golden/make_ref_fixtures.py

/opt/conda/bin/python3.9 carries the real `skimage` (0.18.3) and `h5py` (3.3.0, libhdf5 1.10) the reference calls.
What is imported from the reference, by path, and executed unmodified:

  l3embedding/audio.py:4-31                pcm2float                      -> ref_preprocess.npz  (A6, all 65536 int16 codes)
  l3embedding/train.py:186                 2*img_as_float(u8).astype('float32')-1 (the expression, evaluated with the real
                                           skimage.img_as_float train.py:14 imports) -> ref_preprocess.npz (A6, all 256 codes)
  l3embedding/train.py:142-205             data_generator / single_epoch_data_generator (batch sequence, blob spill-over,
                                           reshuffled passes, start_batch_idx skip, validation restart) -> ref_feed.npz (F3)
  l3embedding/train.py:208-216             get_restart_info               -> ref_feed.npz (F4 resume)
  data/avc/sample.py:565-568               write_to_h5 (gzip HDF5 blobs, real h5py) -> ref_blobs/*.h5 (F3: h5lite must read them)
  data/usc/vggish/mel_features.py:48-68    periodic_hann                  -> ref_dsp.npz (the window kapre folds into its DFT kernels)
  data/usc/vggish/mel_features.py:71-97    stft_magnitude (frame + periodic Hann + rfft magnitude)
                                           -> ref_dsp.npz (== kapre Spectrogram(n_dft=512, n_hop=242, padding='valid',
                                           power 1.0) of audio_model.py:39-40: the cnn_L3_orig front-end before its log)
  data/usc/vggish/mel_features.py:100-111  hertz_to_mel (HTK formula)      -> ref_dsp.npz

train.py and sample.py import, at module level, packages that are neither installed nor used by the functions above
(keras, tensorflow, kapre, pescador, git, gsheets, googleapiclient, skvideo, soundfile, tqdm).  Those imports are
satisfied by inert placeholder modules so that the *unrelated* pure numpy/h5py/skimage functions can run; nothing that
touches a placeholder is executed and nothing numeric comes from one.  Keras/TF/kapre arithmetic itself cannot be
produced here (those rows stay restatement-only).

Only the .npz / .h5 data written here travels; no reference source is copied.
"""
import csv
import importlib.abc
import importlib.machinery
import os
import shutil
import sys
import tempfile
from unittest import mock

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
PLACEHOLDERS = ('git', 'keras', 'pescador', 'gsheets', 'googleapiclient', 'kapre', 'tensorflow', 'skvideo',
                'soundfile', 'tqdm', 'resampy', 'oauth2client', 'httplib2', 'apiclient')


class _Placeholder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split('.')[0] in PLACEHOLDERS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__path__, m.__spec__, m.__name__, m.__loader__ = [], spec, spec.name, self
        return m

    def exec_module(self, module):
        pass


def import_reference():
    sys.meta_path.insert(0, _Placeholder())
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, 'data', 'usc', 'vggish'))
    import scipy
    if not hasattr(scipy, 'misc'):
        import scipy.misc  # noqa: F401  (sample.py:11 names it; unused here)
    import l3embedding.audio as ref_audio
    import l3embedding.train as ref_train
    import mel_features as ref_mel
    import data.avc.sample as ref_sample
    return ref_audio, ref_train, ref_mel, ref_sample


def preprocess_fixture(ref_audio, ref_train):
    pcm = np.arange(-32768, 32768, dtype=np.int16)
    pcm_f32 = ref_audio.pcm2float(pcm, dtype='float32')                         # audio.py:4-31 as called at train.py:189
    u8 = np.arange(256, dtype=np.uint8)
    u8_f32 = 2 * ref_train.img_as_float(u8).astype('float32') - 1               # train.py:186, real skimage
    assert pcm_f32.dtype == np.float32 and u8_f32.dtype == np.float32
    np.savez_compressed(os.path.join(HERE, 'ref_preprocess.npz'), pcm_i16=pcm, pcm_f32=pcm_f32, u8=u8, u8_f32=u8_f32)
    print('ref_preprocess.npz: pcm2float over', pcm.size, 'codes; img_as_float scaling over', u8.size)


def dsp_fixture(ref_audio, ref_mel):
    rng = np.random.RandomState(20180123)
    pcm = rng.randint(-32768, 32768, size=48000).astype(np.int16)
    # a decaying chirp on top of the noise so that the spectrum is not flat
    t = np.arange(48000) / 48000.0
    pcm = np.clip(0.25 * pcm + 12000 * np.sin(2 * np.pi * (200 + 4000 * t) * t) * np.exp(-2 * t), -32768, 32767).astype(np.int16)
    sig = ref_audio.pcm2float(pcm, dtype='float32').astype(np.float64)
    mag = ref_mel.stft_magnitude(sig, fft_length=512, hop_length=242, window_length=512)     # (197, 257)
    assert mag.shape == (197, 257), mag.shape
    hz = np.concatenate([np.linspace(0.0, 24000.0, 1025), [20.0, 700.0, 1000.0, 7600.0]])
    np.savez_compressed(os.path.join(HERE, 'ref_dsp.npz'),
                        hann2048=ref_mel.periodic_hann(2048), hann512=ref_mel.periodic_hann(512),
                        hann480=ref_mel.periodic_hann(480),
                        hz=hz, mel_of_hz=ref_mel.hertz_to_mel(hz),
                        stft_pcm_i16=pcm, stft_mag_512_242=mag)
    print('ref_dsp.npz: periodic_hann 2048/512/480, hertz_to_mel over', hz.size, 'points, stft_magnitude', mag.shape)


BLOB_ROWS = (5, 7, 4)
FEED_BATCH = 6


def feed_fixture(ref_train, ref_sample):
    blob_dir = os.path.join(HERE, 'ref_blobs')
    shutil.rmtree(blob_dir, ignore_errors=True)
    os.makedirs(blob_dir)
    rng = np.random.RandomState(7)
    row = 0
    for i, n in enumerate(BLOB_ROWS):
        lab = rng.randint(0, 2, size=n)
        batch = {                                                               # schema of sample.py:371-386
            'video': rng.randint(0, 256, size=(n, 6, 6, 3)).astype(np.uint8),
            'audio': rng.randint(-32768, 32768, size=(n, 1, 32)).astype(np.int16),
            'label': np.stack([lab, 1 - lab], axis=1),
            'audio_start_sample_idx': np.arange(row, row + n),                  # a metadata key the feed must not deliver
        }
        row += n
        ref_sample.write_to_h5(os.path.join(blob_dir, 'blob_%d.h5' % i), batch)  # sample.py:565-568 (gzip)
    order = os.listdir(blob_dir)                                                # train.py:154 iterates this order as listed

    def take(gen, n):
        out = []
        for _ in range(n):
            b = next(gen)
            assert sorted(b.keys()) == ['audio', 'label', 'video']
            out.append(b)
        return out

    def pack(prefix, batches, dst):
        dst[prefix + '_video'] = np.stack([b['video'] for b in batches])
        dst[prefix + '_audio'] = np.stack([b['audio'] for b in batches])
        dst[prefix + '_label'] = np.stack([b['label'] for b in batches])

    out = {'listdir_order': np.array(order), 'blob_rows': np.array(BLOB_ROWS), 'batch_size': np.array(FEED_BATCH)}
    # 16 rows per pass, batch 6: 14 batches = 84 rows = 5.25 passes (4 reshuffles)
    pack('train', take(ref_train.data_generator(blob_dir, batch_size=FEED_BATCH, random_state=20180123), 14), out)
    pack('resume', take(ref_train.data_generator(blob_dir, batch_size=FEED_BATCH, random_state=20180123,
                                                 start_batch_idx=5), 6), out)
    out['resume_start_batch_idx'] = np.array(5)
    pack('seed99', take(ref_train.data_generator(blob_dir, batch_size=FEED_BATCH, random_state=99), 9), out)
    pack('valid', take(ref_train.single_epoch_data_generator(blob_dir, 2, batch_size=FEED_BATCH, random_state=20180123), 5), out)
    out['valid_epoch_size'] = np.array(2)

    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, 'history_csvlog.csv')
    with open(path, 'w', newline='') as fh:                                     # keras CSVLogger layout (train.py:361)
        w = csv.writer(fh)
        w.writerow(['epoch', 'acc', 'loss', 'val_acc', 'val_loss'])
        w.writerow([0, 0.5, 0.9, 0.515625, 0.8125])
        w.writerow([1, 0.625, 0.75, 0.59375, 0.703125])
    out['restart_csv'] = np.array(open(path).read())
    out['restart_info'] = np.array(ref_train.get_restart_info(path), dtype=np.float64)       # train.py:208-216
    shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(HERE, 'ref_feed.npz'), **out)
    print('ref_feed.npz + ref_blobs/: listdir order', order)


if __name__ == '__main__':
    ref_audio, ref_train, ref_mel, ref_sample = import_reference()
    preprocess_fixture(ref_audio, ref_train)
    dsp_fixture(ref_audio, ref_mel)
    feed_fixture(ref_train, ref_sample)
