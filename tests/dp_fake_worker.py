"""Worker of test_parity_gpu.py::test_dp_event_ordering_with_a_fake_collective.  Runs in its own process because
csrc/comm.hip binds its collective library once per process: here L3_RCCL_LIB names tests/fake_rccl/libfake_rccl.so,
whose all-reduce multiplies by the world size behind a spinning kernel (see fake_rccl.hip)."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def overlap(mt, B, steps, world):
    """Does a slow collective hide behind backward?  The double's all-reduce takes FAKE_RCCL_DELAY_US per bucket on the
    communicator stream: time `steps` data-parallel steps against the same number of plain steps on one engine."""
    import time
    from l3embedding_amd import _lib
    from oracle import l3_oracle as o
    v, a, l = o.synthetic_batch(B, seed=71)
    e = _lib.Engine(mt, B, seed=5, global_batch=world * B)
    e.set_param('dense_2/kernel', e.get_param('dense_2/kernel', (128, 2)) / np.float32(64))      # live loss gradients (bench.py live_head)
    e.comm_init(_lib.comm_unique_id(), world, 0)
    e.upload_batch(v, a, l)

    def run(step):
        for _ in range(5):
            step(1e-5)
        e.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(1e-5)
        e.sync()
        return 1e3 * (time.perf_counter() - t0) / steps
    plain = [run(e.step_resident) for _ in range(2)]
    dp = [run(e.step_dp) for _ in range(2)]
    plain2 = run(e.step_resident)
    e.comm_timing(True)
    for _ in range(5):
        e.step_dp(1e-5)
    ct = e.comm_timing_read()
    e.comm_timing(False)
    out = {'plain_ms': min(plain + [plain2]), 'dp_ms': min(dp), 'delay_us': int(os.environ.get('FAKE_RCCL_DELAY_US', '300')),
           'buckets': e.bucket_count(), 'comm_timing': ct}
    e.comm_destroy()
    e.close()
    print('RESULT ' + json.dumps(out))


def moving(mt, B, steps, world, zero_debias):
    """BatchNorm moving statistics of the data-parallel step (l3_config.dp_moving): with the double's all-gather every one of the
    `world` replicas holds this rank's batch statistics, so the engine must apply `world` moving-average updates per step."""
    from l3embedding_amd import _lib
    from oracle import l3_oracle as o
    out = {}
    for mode in ('replicas', 'rank_local'):
        e = _lib.Engine(mt, B, seed=5, global_batch=world * B, dp_moving=mode, bn_zero_debias=bool(zero_debias))
        e.comm_init(_lib.comm_unique_id(), world, 0)
        for k in range(steps):
            v, a, l = o.synthetic_batch(B, seed=71 + k)        # another batch every step: the statistics must move
            e.upload_batch(v, a, l)
            e.step_dp(0.0)             # frozen weights: the statistics depend on the data alone (see the test)
        W = e.get_params()
        out[mode] = {k: W[k].tolist() for k in W if 'moving_' in k}
        out[mode + '_steps'] = list(e.optimizer_steps())
        _, lg = e.forward(v, a, training=False)            # inference through the moving statistics
        out[mode + '_logits'] = lg.tolist()
        e.comm_destroy()
        e.close()
    print('RESULT ' + json.dumps(out))


def disagree():
    """l3_comm_init compares the ranks' bucket order, bucket count, dp_moving and model / precision (a MAX all-reduce of x and -x).  The
    double plays a peer that reports another value (FAKE_RCCL_MAX_PEER): initialisation must fail with a message that names it."""
    from l3embedding_amd import _lib
    e = _lib.Engine('tiny_L3', 2, seed=5, global_batch=4)
    try:
        e.comm_init(_lib.comm_unique_id(), 2, 0)
        out = {'error': None}
    except _lib.L3Error as exc:
        out = {'error': str(exc)}
    # the engine is usable afterwards: no communicator was left behind
    ok = True
    try:
        e.step_dp(1e-4)
        ok = False
    except _lib.L3Error:
        pass
    out['no_comm_left'] = ok
    e.close()
    print('RESULT ' + json.dumps(out))


def main():
    if sys.argv[1] == 'disagree':
        return disagree()
    if sys.argv[1] == 'overlap':
        return overlap(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    if sys.argv[1] == 'moving':
        return moving(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
    mt, B, steps, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    from l3embedding_amd import _lib
    from oracle import l3_oracle as o
    v, a, l = o.synthetic_batch(B, seed=71)
    # plain single-GPU step at global batch B
    e_ref = _lib.Engine(mt, B, seed=5)
    # data-parallel engine: "world" ranks of batch B each; the fake collective sums world copies of this rank's data
    # (rank-local moving statistics: this comparison is about ORDER, every tensor of the plain step bit for bit; the `world` updates
    # per step of the default dp_moving have their own test, `moving` above)
    e_dp = _lib.Engine(mt, B, seed=5, global_batch=world * B, dp_moving='rank_local')
    e_dp.set_params(e_ref.get_params())
    e_dp.comm_init(_lib.comm_unique_id(), world, 0)
    info = e_dp.comm_info()
    assert 'fake_rccl' in info['library'], info
    e_ref.upload_batch(v, a, l)
    e_dp.upload_batch(v, a, l)
    out = {'library': info['library'], 'param_mismatch': []}
    for _ in range(steps):
        e_dp.step_dp(1e-5)              # small steps: at 1e-3 this 2-pair batch is fitted after two steps, the clipped
        e_ref.step_resident(1e-5)       # cross-entropy (train.py:282-284 semantics) then has a zero gradient everywhere
    Ga, Gb = e_dp.get_grads(), e_ref.get_grads()
    out['grad_mismatch'] = [k for k in Gb if not np.array_equal(Ga[k], Gb[k])]
    out['grad_nonzero'] = int(sum(1 for k in Gb if np.any(Gb[k] != 0)))
    Wa, Wb = e_dp.get_params(), e_ref.get_params()
    out['param_mismatch'] = [k for k in Wb if not np.array_equal(Wa[k], Wb[k])]
    out['n_tensors'] = len(Wb)
    out['host_sum'] = e_dp.comm_allreduce([1.5, -2.0], 'sum')
    out['host_max'] = e_dp.comm_allreduce([1.5, -2.0], 'max')
    fake = ctypes.CDLL(info['library'])
    fake.fake_rccl_launches.restype = ctypes.c_long
    out['collectives'] = int(fake.fake_rccl_launches())
    out['buckets'] = e_dp.bucket_count()
    e_dp.comm_destroy()
    e_dp.close()
    e_ref.close()
    print('RESULT ' + json.dumps(out))


if __name__ == '__main__':
    main()
