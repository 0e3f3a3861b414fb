"""Parity against data computed by the REFERENCE's own code (tests/golden/make_ref_fixtures.py ran
/root/reference's pcm2float, train.py's img_as_float expression with the real skimage, data_generator,
single_epoch_data_generator, get_restart_info, write_to_h5, and the vggish periodic_hann / stft_magnitude /
hertz_to_mel).  These are the rows of SURVEY.md 8 that are *reference-pinned*; see DESIGN.md section 2."""
import os
import shutil

import numpy as np
import pytest

from oracle import l3_oracle as o

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _npz(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


# ---------------------------------------------------------------- A6: input scalings (bit-exact)
def test_oracle_pcm2float_equals_reference_on_every_int16_code():
    f = _npz('ref_preprocess.npz')
    got = o.pcm2float(f['pcm_i16'], np.float32)
    assert got.dtype == np.float32 and np.array_equal(got, f['pcm_f32'])
    assert f['pcm_f32'][0] == -1.0 and f['pcm_f32'][32768] == 0.0 and f['pcm_f32'][-1] == np.float32(32767 / 32768)


def test_oracle_video_scaling_equals_reference_on_every_uint8_code():
    f = _npz('ref_preprocess.npz')
    got = o.preprocess_video(f['u8'])
    assert got.dtype == np.float32 and np.array_equal(got, f['u8_f32'])
    assert f['u8_f32'][0] == -1.0 and f['u8_f32'][255] == 1.0


@pytest.mark.gpu
def test_gpu_preprocess_equals_reference_fixture(gpu_required):
    """l3_op_preprocess (the kernels l3_upload_batch_raw / l3_stage_batch_raw run) against the reference-made table:
    every uint8 code and every int16 code, bit for bit."""
    from l3embedding_amd import _lib
    f = _npz('ref_preprocess.npz')
    vo, ao = _lib.op_preprocess(f['u8'], f['pcm_i16'])
    assert np.array_equal(vo, f['u8_f32'])
    assert np.array_equal(ao, f['pcm_f32'])


# ---------------------------------------------------------------- A2/A8: window, HTK mel scale, 'valid' STFT magnitude
def test_dft_window_is_the_reference_periodic_hann():
    f = _npz('ref_dsp.npz')
    for n in (2048, 512):
        real, imag = o.stft_kernels(n)
        # bin 0 of the real kernels is the window itself (cos 0 = 1), stored as float32 like K.floatx()
        assert np.array_equal(real[:, 0], f['hann%d' % n].astype(np.float32))
        # and the whole kernel is window x DFT basis
        k, t = np.arange(n // 2 + 1), np.arange(n)
        ang = 2 * np.pi * np.outer(t, k) / n
        assert np.abs(real - f['hann%d' % n][:, None] * np.cos(ang)).max() < 1e-6
        assert np.abs(imag + f['hann%d' % n][:, None] * np.sin(ang)).max() < 1e-6


def test_htk_mel_scale_matches_reference_formula():
    """mel_features.py:100-111 writes the HTK scale as 1127 ln(1 + f/700); librosa/kapre (and the oracle) as
    2595 log10(1 + f/700).  2595 / ln 10 = 1127.0105, so the two differ by a constant factor that cancels in a filter
    bank built from mel-equispaced edges.  Pinned: exact proportionality, and the band edges it implies."""
    f = _npz('ref_dsp.npz')
    mine, ref = o._hz_to_mel_htk(f['hz']), f['mel_of_hz']
    nz = ref > 0
    ratio = mine[nz] / ref[nz]
    assert np.abs(ratio - 2595.0 / (1127.0 * np.log(10.0))).max() < 1e-12
    # band edges of the 256-filter bank from the reference's scale: linspace in mel, mapped back with the inverse formula
    edges_ref = 700.0 * (np.exp(np.linspace(ref[0], ref[1024], 258) / 1127.0) - 1.0)
    edges_mine = o._mel_to_hz_htk(np.linspace(mine[0], mine[1024], 258))
    assert np.abs(edges_mine - edges_ref).max() < 1e-8 * 24000


def test_orig_frontend_magnitude_equals_reference_stft_magnitude():
    """audio_model.py:39-40 Spectrogram(n_dft=512, n_hop=242, power 1.0, 'valid') == |rfft(frame * periodic_hann)| of the
    reference's own mel_features.stft_magnitude (same frames: 1 + (48000-512)//242 = 197)."""
    f = _npz('ref_dsp.npz')
    audio = o.pcm2float(f['stft_pcm_i16'], np.float32).reshape(1, 1, -1)
    cfg = dict(o.FRONTENDS['orig'], loglambda=False)
    o.FRONTENDS['_orig_mag'] = cfg
    try:
        mag = o.frontend_forward('_orig_mag', audio, o.frontend_constants('orig'), dtype=np.float64)[0, :, :, 0]
    finally:
        del o.FRONTENDS['_orig_mag']
    ref = f['stft_mag_512_242'].T                                   # (257, 197)
    assert mag.shape == ref.shape == (257, 197)
    # the oracle holds kapre's kernels in float32 (K.floatx()): 512 taps of relative rounding 6e-8 on values up to ~60
    assert np.abs(mag - ref).max() < 2e-5 * ref.max()
    # and the complete orig front-end (log(max(x,1e-12))/5, audio_model.py:43) from the reference magnitudes
    full = o.frontend_forward('orig', audio, dtype=np.float64)[0, :, :, 0]
    assert np.abs(full - np.log(np.maximum(ref, 1e-12)) / 5.0).max() < 1e-4


@pytest.mark.gpu
def test_gpu_orig_frontend_equals_reference_stft(gpu_required):
    from l3embedding_amd import _lib
    f = _npz('ref_dsp.npz')
    audio = np.repeat(o.pcm2float(f['stft_pcm_i16'], np.float32).reshape(1, 1, -1), 2, axis=0)
    got = _lib.op_frontend('cnn_L3_orig', audio)
    ref = np.log(np.maximum(f['stft_mag_512_242'].T, 1e-12)) / 5.0
    assert got.shape == (2, 257, 197, 1)
    assert np.abs(got[0, :, :, 0] - ref).max() < 5e-4 and np.array_equal(got[0], got[1])


# ---------------------------------------------------------------- F3/F4: batch feed against the reference generator
@pytest.fixture
def ref_blob_dir(tmp_path, monkeypatch):
    f = _npz('ref_feed.npz')
    d = tmp_path / 'blobs'
    shutil.copytree(os.path.join(GOLD, 'ref_blobs'), str(d))
    order = [str(x) for x in f['listdir_order']]
    real_listdir = os.listdir
    # the reference iterates os.listdir() as listed (train.py:154); the order is the file system's, so replay the recorded one
    monkeypatch.setattr(os, 'listdir', lambda p='.': list(order) if os.path.abspath(p) == str(d) else real_listdir(p))
    return str(d), f


def _check(feed, f, prefix, n):
    for i in range(n):
        b = next(feed)
        assert sorted(b.keys()) == ['audio', 'label', 'video']
        # stored dtypes are delivered; the reference's scalings run afterwards (on the GPU in the product)
        assert np.array_equal(o.preprocess_video(b['video']), f[prefix + '_video'][i]), (prefix, i)
        assert np.array_equal(o.pcm2float(b['audio'], np.float32), f[prefix + '_audio'][i]), (prefix, i)
        assert np.array_equal(b['label'], f[prefix + '_label'][i]), (prefix, i)


def test_blobfeed_replays_reference_data_generator(ref_blob_dir):
    """h5lite reads blobs written by the reference's write_to_h5 (real h5py, gzip) and BlobFeed delivers exactly the
    batches l3embedding.train.data_generator yielded: spill-over across blobs, four reshuffled passes, another seed,
    the start_batch_idx resume skip, and the restarting validation generator."""
    from l3embedding_amd.blobfeed import BlobFeed, RestartingFeed
    d, f = ref_blob_dir
    bs = int(f['batch_size'])
    _check(BlobFeed(d, batch_size=bs, random_state=20180123), f, 'train', f['train_label'].shape[0])
    _check(BlobFeed(d, batch_size=bs, random_state=99), f, 'seed99', f['seed99_label'].shape[0])
    _check(BlobFeed(d, batch_size=bs, random_state=20180123, start_batch_idx=int(f['resume_start_batch_idx'])),
           f, 'resume', f['resume_label'].shape[0])
    _check(RestartingFeed(lambda: BlobFeed(d, batch_size=bs, random_state=20180123), int(f['valid_epoch_size'])),
           f, 'valid', f['valid_label'].shape[0])


def test_sharded_blobfeed_tiles_the_reference_batches(ref_blob_dir):
    from l3embedding_amd.blobfeed import BlobFeed
    d, f = ref_blob_dir
    bs = int(f['batch_size'])
    feeds = [BlobFeed(d, batch_size=bs, random_state=20180123, rank=r, world=4) for r in range(4)]
    for i in range(f['train_label'].shape[0]):
        parts = [next(fd) for fd in feeds]
        assert [len(p['label']) for p in parts] == [1, 1, 1, 3]           # training_utils.py:121-133: remainder on the last
        assert np.array_equal(np.concatenate([p['label'] for p in parts]), f['train_label'][i])
        assert np.array_equal(o.preprocess_video(np.concatenate([p['video'] for p in parts])), f['train_video'][i])


def test_restart_info_equals_reference(tmp_path):
    from l3embedding_amd.train import get_restart_info
    f = _npz('ref_feed.npz')
    p = tmp_path / 'history_csvlog.csv'
    p.write_text(str(f['restart_csv']))
    got = get_restart_info(str(p))
    assert tuple(got) == tuple(f['restart_info']) and isinstance(got[0], int)
