"""Data parallelism for the L3 AVC training step: one process per GPU, RCCL all-reduce.

Stands in for l3embedding/training_utils.py:21-170 (`multi_gpu_model`, the
reference's only parallelism strategy: single-process in-graph replication with a CPU
concat and an implicit gradient AddN).  Semantics kept:
  * the global batch is split by rank with `get_slice` arithmetic
    (training_utils.py:121-133: step = B // gpus, last replica takes the remainder);
  * the optimiser sees the gradient of the MEAN loss over the concatenated batch: every
    rank scales its loss gradient by 1/global_batch and the ranks' gradients are SUMMED;
  * BatchNorm batch statistics are per replica (no sync-BN), as in the reference; the MOVING statistics receive one update
    per replica and step, in replica order, on every rank alike (the reference calls its one template model once per
    replica, training_utils.py:141-157, so every BatchNormalization updates its shared moving variables `gpus` times);
  * every rank holds identical weights (same init seed, same reduced gradients).
MI355X-native differences: ranks are processes (torch.distributed, backend "nccl" ==
RCCL over xGMI), and the fp32 gradient arena is reduced in buckets that become ready
head -> block4 -> ... -> block1 while backward is still running; the collectives run on
RCCL's own side stream, ordered against the engine's stream by events.
"""
import os

import numpy as np


def get_slice_bounds(batch_size, parts, i):
    """[start, stop) of replica `i` -- training_utils.py:121-133."""
    step = batch_size // parts
    start = step * i
    size = batch_size - step * i if i == parts - 1 else step
    return start, start + size


class _DevArray(object):
    """Exposes a raw device pointer through __cuda_array_interface__ (zero-copy)."""

    def __init__(self, ptr, numel):
        self.__cuda_array_interface__ = {
            'shape': (int(numel),), 'typestr': '<f4', 'data': (int(ptr), False), 'version': 2, 'strides': None}


class GradientAverager(object):
    """Bucketed SUM all-reduce of a flat gradient buffer over torch.distributed.

    `flat` is a torch tensor (CUDA for RCCL, CPU for the gloo tests) viewing the whole
    gradient arena; `ranges` lists (offset, numel) per bucket in ready order.
    """

    def __init__(self, flat, ranges, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.flat = flat
        self.views = [flat.narrow(0, int(o), int(n)) for o, n in ranges]
        self.group = group
        self.pending = []

    def reduce_bucket(self, k):
        """Launch the all-reduce of bucket k; returns immediately (async work handle kept)."""
        if self.views[k].numel() == 0:
            return
        self.pending.append(self.dist.all_reduce(self.views[k], op=self.dist.ReduceOp.SUM,
                                                 group=self.group, async_op=True))

    def wait(self):
        """Make the compute stream (or the host, for gloo) wait for every pending bucket."""
        for w in self.pending:
            w.wait()
        self.pending = []


class ReplicaStatsGather(object):
    """All-gather of the ranks' packed BatchNorm batch statistics over torch.distributed (l3_config.dp_moving = replicas).

    `send`: this rank's (n,) tensor; `recv`: the (world * n,) tensor every rank's statistics arrive in, rank-major -- CUDA tensors
    viewing the engine's buffers (l3_bn_stats_pack_dev / l3_bn_stats_replicas_dev) for RCCL, CPU tensors in the gloo tests."""

    def __init__(self, send, recv, world, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.send = send
        self.recv = recv
        self.slots = list(recv.chunk(int(world)))
        assert len(self.slots) == int(world) and all(s.numel() == send.numel() for s in self.slots)
        self.group = group
        self.rank = dist.get_rank(group)
        self.pending = None

    def launch(self):
        # an all-gather written as a SUM all-reduce of a buffer that is zero outside this rank's slot (x + 0 is exact): gloo, the
        # backend of the CPU / one-GPU tests, has no all_gather for CUDA tensors, and 7.6 k floats per rank are not worth a second path
        self.recv.zero_()
        self.slots[self.rank].copy_(self.send)
        self.pending = self.dist.all_reduce(self.recv, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None


class DataParallelTrainer(object):
    """Drives one rank's engine through a data-parallel step.

    engine: l3embedding_amd._lib.Engine created with batch = this rank's shard size and
            global_batch = the batch over all ranks, on the torch current stream.
    """

    def __init__(self, engine, device_index, world_size, rank, group=None, stream=None, flat=None):
        """`stream`: the torch.cuda.Stream the engine was created on (engine launches and the
        collectives' stream dependencies must refer to the same stream).  It must be a real
        side stream: the legacy default stream has handle 0, which the C ABI reads as "make
        your own stream", and the all-reduce would then not be ordered after backward."""
        import torch
        self.torch = torch
        self.engine = engine
        self.world = int(world_size)
        self.rank = int(rank)
        self.stream = stream
        self.staged = None
        if flat is not None:
            # host-side double of the gradient arena (world_size-2 gloo tests): no device, no stream
            self.flat = flat
        else:
            if self.world > 1 and stream is None:
                raise ValueError('DataParallelTrainer needs the torch.cuda.Stream the engine runs on when world_size > 1')
            ptr, n = engine.grad_arena()
            dev = 'cuda:%d' % device_index
            try:
                flat = torch.as_tensor(_DevArray(ptr, n), device=dev)
                if flat.data_ptr() != ptr:
                    raise RuntimeError('copy made')
            except Exception:
                # fall back to a torch-owned staging buffer (2 extra D2D copies of 38 MB per step)
                flat = torch.empty(n, dtype=torch.float32, device=dev)
                self.staged = (ptr, n)
            self.flat = flat
        nb = engine.bucket_count()
        self.ranges = [engine.bucket_range(b) for b in range(nb)]
        self.avg = GradientAverager(self.flat, self.ranges, group)
        self.group = group
        self.stats = None            # ReplicaStatsGather over the engine's buffers, built at the first step

    def _wrap(self, ptr, n):
        """Zero-copy torch view of `n` floats of device memory at `ptr`."""
        t = self.torch.as_tensor(_DevArray(ptr, n), device=self.flat.device)
        if t.data_ptr() != ptr:
            raise RuntimeError('torch copied the BatchNorm statistics buffer instead of viewing it')
        return t

    def _gather_replica_stats(self):
        """Pack this rank's BatchNorm batch statistics behind the forward pass and all-gather them; the update then applies one
        moving-average step per replica (l3_bn_stats_*_dev).  Engines / doubles without the hooks keep rank-local statistics."""
        e = self.engine
        if getattr(e, 'dp_moving', 'rank_local') != 'replicas' or not hasattr(e, 'bn_stats_pack'):
            return
        ptr, n = e.bn_stats_pack()
        if n == 0:
            return
        gptr = e.bn_stats_replicas(self.world)
        if self.stats is None or self.stats.send.data_ptr() != ptr or self.stats.slots[0].data_ptr() != gptr:
            self.stats = ReplicaStatsGather(self._wrap(ptr, n), self._wrap(gptr, n * self.world), self.world, self.group)
        self.stats.launch()

    _hip = None

    @classmethod
    def _hiprt(cls):
        if cls._hip is None:
            import ctypes
            cls._hip = ctypes.CDLL('libamdhip64.so')
        return cls._hip

    def _stage_in(self, k):
        if self.staged is None:
            return
        import ctypes
        o, n = self.ranges[k]
        hip = self._hiprt()
        hip.hipMemcpyAsync(ctypes.c_void_p(self.flat.data_ptr() + 4 * o), ctypes.c_void_p(self.staged[0] + 4 * o),
                           ctypes.c_size_t(4 * n), 3, ctypes.c_void_p(self.stream.cuda_stream))

    def _stage_out(self):
        if self.staged is None:
            return
        import ctypes
        hip = self._hiprt()
        hip.hipMemcpyAsync(ctypes.c_void_p(self.staged[0]), ctypes.c_void_p(self.flat.data_ptr()),
                           ctypes.c_size_t(4 * self.staged[1]), 3, ctypes.c_void_p(self.stream.cuda_stream))

    def step(self, lr):
        """forward -> backward with overlapped bucket all-reduce -> Adam.  Inputs must
        already be resident (engine.upload_batch*)."""
        e = self.engine
        if self.world <= 1:
            e.step_forward(True)
            for b in range(1, len(self.ranges)):
                e.step_backward_bucket(b)
            e.step_update(lr, 1.0)
            return
        # torch orders each collective after the work already queued on the *current* stream and
        # work.wait() makes the current stream wait for RCCL's side stream: keep the engine's
        # stream current for the whole step
        import contextlib
        ctx = self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()
        with ctx:
            e.step_forward(True)
            self._stage_in(0)
            self.avg.reduce_bucket(0)
            self._gather_replica_stats()
            for b in range(1, len(self.ranges)):
                e.step_backward_bucket(b)
                self._stage_in(b)
                self.avg.reduce_bucket(b)
            self.avg.wait()
            if self.stats is not None:
                self.stats.wait()
            self._stage_out()
            # loss gradients were scaled by 1/global_batch, so the SUM is already the mean
            e.step_update(lr, 1.0)


def share_unique_id(rank, world):
    """The 128-byte ncclUniqueId of the job on every rank: rank 0 asks the library for one
    (`l3_comm_unique_id`) and the bytes travel over whatever rendezvous the launcher already provides --
    the initialised torch.distributed process group if there is one, else the env:// key-value store that
    `python -m torch.distributed.run` sets up (MASTER_ADDR / MASTER_PORT), without creating a process group."""
    from . import _lib
    if world == 1:
        return _lib.comm_unique_id()
    _lib.require_single_hip_runtime('exchanging the RCCL unique id over torch.distributed')
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        box = [_lib.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    # One env:// store per process, kept for its lifetime: a second rendezvous would try to re-bind the
    # master port on rank 0.  Every communicator of the process gets its own key (ranks create their
    # communicators in the same order, so the per-process counter agrees across ranks): a later engine can
    # never read the id of an earlier communicator.
    store = _env_store(rank, world)
    n = share_unique_id._count = getattr(share_unique_id, '_count', 0) + 1
    key = 'l3hip/nccl_unique_id/%s/%d' % (os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), n)
    if rank == 0:
        store.set(key, _lib.comm_unique_id())
    return bytes(store.get(key))


def _env_store(rank, world):
    store = getattr(_env_store, '_store', None)
    if store is None:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        from torch.distributed import rendezvous
        store, _, _ = next(rendezvous('env://', rank=rank, world_size=world))
        _env_store._store = store
    return store


class NativeDataParallelTrainer(object):
    """Data-parallel step with the collectives INSIDE libl3hip.so (`l3_comm_init` + `l3_step_dp`): the bucketed
    ncclAllReduce calls are enqueued by the library on its communicator stream behind per-bucket events --
    no Python between backward and the all-reduce.  `DataParallelTrainer` (torch.distributed) is the
    test double of this path."""

    reduces_results = True      # Engine.results_enqueue(reduce=True) sums the step's loss / accuracy over the ranks on the device

    def __init__(self, engine, world_size, rank, unique_id=None):
        self.engine, self.world, self.rank = engine, int(world_size), int(rank)
        engine.comm_init(unique_id if unique_id is not None else share_unique_id(self.rank, self.world), self.world, self.rank)

    def step(self, lr):
        self.engine.step_dp(lr)

    def allreduce(self, values, op='sum'):
        return self.engine.comm_allreduce(values, op)

    def barrier(self):
        self.engine.sync()
        self.engine.comm_allreduce([0.0])


def available_devices():
    """Device names in the reference's notation ('/cpu:0', '/gpu:0', ...; training_utils.py:12-18,107-109):
    the AMD GPUs this node offers the job.  One process per GPU: a rank that was handed a single visible
    device still counts its node's ranks (LOCAL_WORLD_SIZE, set by torch.distributed.run)."""
    from . import _lib
    try:
        n = max(0, int(_lib.load().l3_device_count()))
    except Exception:
        n = 0
    n = max(n, int(os.environ.get('LOCAL_WORLD_SIZE', '0') or 0) if n else 0)
    return ['/cpu:0'] + ['/gpu:%d' % i for i in range(n)]


def check_gpus_available(gpus):
    """The device check of training_utils.py:100-119, with its message."""
    target_devices = ['/cpu:0'] + ['/gpu:%d' % i for i in range(gpus)]
    have = available_devices()
    if any(d not in have for d in target_devices):
        raise ValueError('To call `multi_gpu_model` with `gpus=%d`, we expect the following devices to be '
                         'available: %s. However this machine only has: %s. Try reducing `gpus`.'
                         % (gpus, target_devices, have))


def multi_gpu_model(model, gpus, validate=True):
    """Reference-compatible entry point (training_utils.py:21): wrap `model` for data-parallel training on
    `gpus` devices, with the reference's argument checks and messages (training_utils.py:99-119).  Here a
    replica is a process, so the wrapper records the replica count and the model shards each fed batch by
    torch.distributed rank.  validate=False (used when only a wrapper-layout weight FILE is being read,
    model.load_model) skips the device check, so a file trained on 8 GPUs converts on a smaller box."""
    if gpus <= 1:
        raise ValueError('For multi-gpu usage to be effective, call `multi_gpu_model` with `gpus >= 2`. '
                         'Received: `gpus=%d`' % gpus)
    if validate:
        check_gpus_available(gpus)
    return model.as_data_parallel(gpus)


def shard_batch(arrays, parts, i):
    """Slice every array of a fed batch for replica i (training_utils.py:141-155)."""
    n = len(arrays[0])
    a, b = get_slice_bounds(n, parts, i)
    return [np.asarray(x)[a:b] for x in arrays]
