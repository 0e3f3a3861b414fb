"""ctypes binding of libl3hip.so (the C ABI declared in include/l3hip.h).

There is deliberately no CPU fallback: if the HIP library is missing or no AMD GPU is
visible the product path raises.
"""
import ctypes as C
import os

import numpy as np

from . import _build

MODEL_IDS = {
    'cnn_L3_orig': 0,
    'tiny_L3': 1,
    'cnn_L3_kapredbinputbn': 2,
    'cnn_L3_melspec1': 3,
    'cnn_L3_melspec2': 4,
}

FAMILIES = ('conv_fwd', 'conv_dgrad', 'conv_wgrad', 'elementwise', 'frontend', 'head', 'adam')


DTYPES = {'f32': 0, 'fp32': 0, 'float32': 0, 'bf16': 1}
FP32_CONV = {'f4x4': 0, 'f2x2': 1, 'f2x2_bf16x6': 2}        # L3_FP32_CONV_*: Winograd F(4x4,3x3) (default, fastest) / F(2x2,3x3) (tightest parity)
DP_MOVING = {'replicas': 0, 'rank_local': 1}       # l3_config.dp_moving (include/l3hip.h L3_DP_MOVING_*)
OP_DTYPES = dict(DTYPES, bf16_stored=2, bf16_stored_out=3)     # L3_OP_BF16_STORED: conv operator entry points only


class L3Config(C.Structure):
    _fields_ = [
        ('struct_size', C.c_int32),
        ('model_type', C.c_int32),
        ('batch', C.c_int32),
        ('global_batch', C.c_int32),
        ('device', C.c_int32),
        ('db_max_scope', C.c_int32),
        ('bn_zero_debias', C.c_int32),
        ('dtype', C.c_int32),
        ('stream', C.c_void_p),
        ('fp32_conv', C.c_int32),
        ('dp_moving', C.c_int32),
    ]


class L3Error(RuntimeError):
    pass


_f32p = C.POINTER(C.c_float)
_lib = None
TORCH_LOADED_FIRST = None     # set by load()

# name -> (restype, argtypes); every symbol include/l3hip.h declares
SIGNATURES = {
    'l3_create': (C.c_int, [C.POINTER(L3Config), C.c_uint64, C.POINTER(C.c_void_p)]),
    'l3_destroy': (None, [C.c_void_p]),
    'l3_last_error': (C.c_char_p, [C.c_void_p]),
    'l3_build_experiments': (C.c_int, []),
    'l3_comm_version': (C.c_int, []),
    'l3_bn_stats_pack_dev': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    'l3_bn_stats_replicas_dev': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    'l3_model_type_from_name': (C.c_int, [C.c_char_p]),
    'l3_device_count': (C.c_int, []),
    'l3_param_count': (C.c_int, [C.c_void_p]),
    'l3_param_info': (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int32),
                                C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    'l3_set_param': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    'l3_get_param': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    'l3_get_grad': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    'l3_reset_optimizer': (C.c_int, [C.c_void_p]),
    'l3_copy_state': (C.c_int, [C.c_void_p, C.c_void_p]),
    'l3_optimizer_steps': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'l3_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'l3_train_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    'l3_eval_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    'l3_upload_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'l3_upload_batch_raw': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'l3_tower_step': (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    'l3_stage_batch_raw': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'l3_step_forward': (C.c_int, [C.c_void_p, C.c_int]),
    'l3_step_bucket_count': (C.c_int, [C.c_void_p]),
    'l3_step_backward_bucket': (C.c_int, [C.c_void_p, C.c_int]),
    'l3_step_update': (C.c_int, [C.c_void_p, C.c_float, C.c_float]),
    'l3_step_resident': (C.c_int, [C.c_void_p, C.c_float]),
    'l3_step_results': (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_void_p]),
    'l3_step_results_enqueue': (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    'l3_step_results_wait': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    'l3_comm_unique_id': (C.c_int, [C.c_void_p]),
    'l3_comm_init': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    'l3_comm_destroy': (C.c_int, [C.c_void_p]),
    'l3_comm_info': (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    'l3_comm_allreduce_host': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]),
    'l3_step_dp': (C.c_int, [C.c_void_p, C.c_float]),
    'l3_comm_timing': (C.c_int, [C.c_void_p, C.c_int]),
    'l3_comm_timing_read': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]),
    'l3_grad_arena_dev': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    'l3_bucket_range': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'l3_embed_audio': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    'l3_embed_vision': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    'l3_embed_dim': (C.c_int64, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    'l3_get_activation': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    'l3_activation_numel': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    'l3_sync': (C.c_int, [C.c_void_p]),
    'l3_profile_enable': (C.c_int, [C.c_void_p, C.c_int]),
    'l3_set_tower_overlap': (C.c_int, [C.c_void_p, C.c_int]),
    'l3_profile_read_executed': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    'l3_profile_read_bytes': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    'l3_profile_read': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_double)]),
    'l3_op_conv2d_fwd': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8),
    'l3_op_conv2d_fwd_dt': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8),
    'l3_op_conv2d_bwd_dt': (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 8),
    'l3_op_conv2d_bwd': (C.c_int, [C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 8),
    'l3_op_bn_relu_fwd': (C.c_int, [C.c_int] + [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_int]),
    'l3_op_bn_relu_bwd': (C.c_int, [C.c_int] + [C.c_void_p] * 10 + [C.c_int64, C.c_int, C.c_int, C.c_int]),
    'l3_op_bn_relu_pool2_fwd': (C.c_int, [C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 7),
    'l3_op_bn_relu_pool2_bwd': (C.c_int, [C.c_int] + [C.c_void_p] * 8 + [C.c_int] * 7),
    'l3_op_maxpool_fwd': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 9),
    'l3_op_maxpool_bwd': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 9),
    'l3_op_frontend': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'l3_op_bn_stats_from_partials': (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_float,
                                               C.c_void_p, C.c_void_p]),
    'l3_op_preprocess': (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
}


def lib_path():
    # developer A/B of two builds on one box (scripts/ab_step.sh); like every other switch only behind L3_DEBUG_KNOBS=1
    if os.environ.get('L3_DEBUG_KNOBS') == '1' and os.environ.get('L3_LIB_PATH'):
        return os.path.abspath(os.environ['L3_LIB_PATH'])
    return _build.LIBPATH


def load():
    """dlopen libl3hip.so and bind every declared symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise L3Error('libl3hip.so not built (%s); run `python -c "import __graft_entry__ as g; g.build()"`. '
                      'There is no CPU fallback.' % path)
    global TORCH_LOADED_FIRST
    import sys
    # PyTorch-ROCm bundles its own libamdhip64 / librccl under the system SONAMEs.  Imported BEFORE this
    # library, its copies satisfy libl3hip's dependencies and the process has one HIP runtime (needed to
    # share streams / device pointers with torch or with the RCCL torch loaded); imported AFTER, the
    # process ends up with two runtimes that cannot see each other's memory.
    TORCH_LOADED_FIRST = 'torch' in sys.modules
    # An engine uses up to four HIP streams (two towers, input staging, RCCL); with HIP's default of 4 hardware queues
    # per process they start sharing queues with each other (and with torch's), and one stream's event waits become
    # false dependencies of another (csrc/comm.hip).  Read by the HIP runtime when it initialises, so it only takes
    # effect if no HIP call was made yet in this process; launchers should export it themselves.
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def experiments_built():
    """True if libl3hip.so carries the measured-and-rejected kernel variants (_build.py L3_BUILD_EXPERIMENTS=1)."""
    return bool(load().l3_build_experiments())


def comm_unique_id():
    """Rank 0: the 128-byte ncclUniqueId to hand to every rank's Engine.comm_init()."""
    buf = C.create_string_buffer(128)
    check(load().l3_comm_unique_id(buf))
    return buf.raw


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def require_single_hip_runtime(what):
    """Raises if torch is about to be (or was) imported after libl3hip in this process."""
    import sys
    if _lib is not None and not TORCH_LOADED_FIRST and 'torch' in sys.modules:
        raise L3Error('%s needs torch and libl3hip to share one HIP runtime: `import torch` before the first engine / '
                      'l3embedding_amd._lib.load() in this process' % what)
    if _lib is not None and not TORCH_LOADED_FIRST and 'torch' not in sys.modules:
        raise L3Error('%s would import torch after libl3hip was loaded (two HIP runtimes in one process): '
                      '`import torch` first' % what)


def check(rc, handle=None):
    if rc != 0:
        msg = load().l3_last_error(handle)
        raise L3Error('libl3hip error %d: %s' % (rc, msg.decode() if msg else '?'))


class Engine(object):
    """Thin RAII wrapper over an l3_engine handle."""

    def __init__(self, model_type, batch, device=0, global_batch=0, db_max_scope='sample',
                 bn_zero_debias=True, seed=20180123, stream=None, dtype='f32', fp32_conv='f4x4', dp_moving='replicas'):
        if model_type not in MODEL_IDS:
            raise ValueError('Invalid model type: "{}"'.format(model_type))
        self.lib = load()
        self.model_type = model_type
        self.batch = int(batch)
        cfg = L3Config()
        cfg.struct_size = C.sizeof(L3Config)
        cfg.model_type = MODEL_IDS[model_type]
        cfg.batch = int(batch)
        cfg.global_batch = int(global_batch)
        cfg.device = int(device)
        cfg.db_max_scope = 0 if db_max_scope == 'sample' else 1
        cfg.bn_zero_debias = 1 if bn_zero_debias else 0
        cfg.stream = stream
        if dtype not in DTYPES:
            raise ValueError('dtype must be one of %s' % sorted(DTYPES))
        cfg.dtype = DTYPES[dtype]
        self.dtype = dtype
        if fp32_conv not in FP32_CONV:
            raise ValueError('fp32_conv must be one of %s' % sorted(FP32_CONV))
        cfg.fp32_conv = FP32_CONV[fp32_conv]
        self.fp32_conv = fp32_conv
        if dp_moving not in DP_MOVING:
            raise ValueError('dp_moving must be one of %s' % sorted(DP_MOVING))
        cfg.dp_moving = DP_MOVING[dp_moving]
        self.dp_moving = dp_moving
        h = C.c_void_p()
        rc = self.lib.l3_create(C.byref(cfg), int(seed), C.byref(h))
        check(rc, None)
        self.h = h
        self._params = None

    def close(self):
        if getattr(self, 'h', None):
            self.lib.l3_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters ---------------------------------------------------------------------
    def param_table(self):
        if self._params is None:
            out = []
            n = self.lib.l3_param_count(self.h)
            buf = C.create_string_buffer(256)
            for i in range(n):
                nd, tr, ne = C.c_int32(), C.c_int32(), C.c_int64()
                shp = (C.c_int64 * 4)()
                check(self.lib.l3_param_info(self.h, i, buf, 256, C.byref(nd), shp, C.byref(tr), C.byref(ne)), self.h)
                out.append((buf.value.decode(), tuple(int(shp[k]) for k in range(nd.value)), bool(tr.value)))
            self._params = out
        return self._params

    def set_param(self, name, value):
        v = _f32(value)
        check(self.lib.l3_set_param(self.h, name.encode(), _ptr(v), v.size), self.h)

    def get_param(self, name, shape):
        out = np.empty(shape, dtype=np.float32)
        check(self.lib.l3_get_param(self.h, name.encode(), _ptr(out), out.size), self.h)
        return out

    def get_grad(self, name, shape):
        out = np.empty(shape, dtype=np.float32)
        check(self.lib.l3_get_grad(self.h, name.encode(), _ptr(out), out.size), self.h)
        return out

    def set_params(self, P):
        for name, shape, _ in self.param_table():
            if name in P:
                self.set_param(name, np.asarray(P[name]).reshape(shape))

    def get_params(self):
        from collections import OrderedDict
        return OrderedDict((n, self.get_param(n, s)) for n, s, _ in self.param_table())

    def get_grads(self):
        from collections import OrderedDict
        return OrderedDict((n, self.get_grad(n, s)) for n, s, t in self.param_table() if t)

    def reset_optimizer(self):
        check(self.lib.l3_reset_optimizer(self.h), self.h)

    def copy_state_from(self, other):
        """Parameters, Adam moments / step and BatchNorm debias accumulators of `other`, device to device."""
        check(self.lib.l3_copy_state(self.h, other.h), self.h)

    def optimizer_steps(self):
        """(Adam iterations, BatchNorm moving-average updates) applied so far."""
        t, b = C.c_int64(), C.c_int64()
        check(self.lib.l3_optimizer_steps(self.h, C.byref(t), C.byref(b)), self.h)
        return t.value, b.value

    # -- steps --------------------------------------------------------------------------
    def forward(self, video, audio, training=False):
        v, a = _f32(video), _f32(audio)
        assert v.shape[0] == self.batch and a.shape[0] == self.batch
        probs = np.empty((self.batch, 2), np.float32)
        logits = np.empty((self.batch, 2), np.float32)
        check(self.lib.l3_forward(self.h, _ptr(v), _ptr(a), int(training), _ptr(probs), _ptr(logits)), self.h)
        return probs, logits

    def train_step(self, video, audio, labels, lr):
        v, a, l = _f32(video), _f32(audio), _f32(labels)
        loss, acc = C.c_float(), C.c_float()
        check(self.lib.l3_train_step(self.h, _ptr(v), _ptr(a), _ptr(l), lr, C.byref(loss), C.byref(acc)), self.h)
        return loss.value, acc.value

    def eval_step(self, video, audio, labels):
        v, a, l = _f32(video), _f32(audio), _f32(labels)
        loss, acc = C.c_float(), C.c_float()
        check(self.lib.l3_eval_step(self.h, _ptr(v), _ptr(a), _ptr(l), C.byref(loss), C.byref(acc)), self.h)
        return loss.value, acc.value

    def upload_batch(self, video=None, audio=None, labels=None):
        v, a, l = _f32(video), _f32(audio), _f32(labels)
        check(self.lib.l3_upload_batch(self.h, _ptr(v), _ptr(a), _ptr(l)), self.h)

    def upload_batch_raw(self, video_u8=None, audio_i16=None, labels_i32=None):
        v = None if video_u8 is None else np.ascontiguousarray(video_u8, dtype=np.uint8)
        a = None if audio_i16 is None else np.ascontiguousarray(audio_i16, dtype=np.int16)
        l = None if labels_i32 is None else np.ascontiguousarray(labels_i32, dtype=np.int32)
        check(self.lib.l3_upload_batch_raw(self.h, _ptr(v), _ptr(a), _ptr(l)), self.h)

    def stage_batch_raw(self, video_u8, audio_i16, labels_i32):
        """Next batch to the device while the current step runs; adopted by the next step_forward()."""
        v = np.ascontiguousarray(video_u8, dtype=np.uint8)
        a = np.ascontiguousarray(audio_i16, dtype=np.int16)
        l = np.ascontiguousarray(labels_i32, dtype=np.int32)
        check(self.lib.l3_stage_batch_raw(self.h, _ptr(v), _ptr(a), _ptr(l)), self.h)

    def tower_step(self, tower, backward=True):
        """One tower alone ('vision' | 'audio') on the resident batch: training-mode forward (+ backward
        from mean(output))."""
        check(self.lib.l3_tower_step(self.h, {'vision': 0, 'audio': 1}[tower], int(backward)), self.h)

    def step_forward(self, training=True):
        check(self.lib.l3_step_forward(self.h, int(training)), self.h)

    def bucket_count(self):
        return self.lib.l3_step_bucket_count(self.h)

    def step_backward_bucket(self, b):
        check(self.lib.l3_step_backward_bucket(self.h, b), self.h)

    def step_update(self, lr, grad_scale=1.0):
        check(self.lib.l3_step_update(self.h, lr, grad_scale), self.h)

    def step_resident(self, lr):
        check(self.lib.l3_step_resident(self.h, lr), self.h)

    def step_results(self, want_probs=False):
        loss, acc = C.c_float(), C.c_float()
        probs = np.empty((self.batch, 2), np.float32) if want_probs else None
        logits = np.empty((self.batch, 2), np.float32) if want_probs else None
        check(self.lib.l3_step_results(self.h, C.byref(loss), C.byref(acc), _ptr(probs), _ptr(logits)), self.h)
        if want_probs:
            return loss.value, acc.value, probs, logits
        return loss.value, acc.value

    def results_enqueue(self, slot, reduce=False):
        """Copies the loss / accuracy sums of the step just enqueued to pinned slot 0 / 1 behind it (no wait).  reduce: summed
        over the ranks of the engine's communicator first (every rank calls it)."""
        check(self.lib.l3_step_results_enqueue(self.h, int(slot), 1 if reduce else 0), self.h)

    def results_wait(self, slot):
        """(loss, acc) of the step whose results went to `slot`; waits for that copy only."""
        loss, acc = C.c_float(), C.c_float()
        check(self.lib.l3_step_results_wait(self.h, int(slot), C.byref(loss), C.byref(acc)), self.h)
        return loss.value, acc.value

    # -- data parallelism through the library's own RCCL communicator -----------------------------------------
    def comm_init(self, unique_id, world, rank):
        """ncclCommInitRank for this engine's GPU; `unique_id` = the 128 bytes rank 0 got from comm_unique_id()."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(self.lib.l3_comm_init(self.h, buf, int(world), int(rank)), self.h)

    def comm_destroy(self):
        check(self.lib.l3_comm_destroy(self.h), self.h)

    def comm_info(self):
        w, r = C.c_int(), C.c_int()
        path = C.create_string_buffer(512)
        check(self.lib.l3_comm_info(self.h, C.byref(w), C.byref(r), path, 512), self.h)
        return dict(world=w.value, rank=r.value, library=path.value.decode())

    def comm_allreduce(self, values, op='sum'):
        """Sum / max of a few host doubles over the ranks (synchronises: also a barrier)."""
        vals = (C.c_double * len(values))(*[float(v) for v in values])
        check(self.lib.l3_comm_allreduce_host(self.h, vals, len(values), {'sum': 0, 'max': 1}[op]), self.h)
        return list(vals)

    def comm_timing(self, on=True):
        """Measurement mode of step_dp (hipEvents around every bucket's all-reduce; every step is waited for)."""
        check(self.lib.l3_comm_timing(self.h, 1 if on else 0), self.h)

    def comm_timing_read(self):
        nb = self.bucket_count()
        ex, sp, st = C.c_double(), C.c_double(), C.c_int()
        bm = (C.c_double * nb)()
        check(self.lib.l3_comm_timing_read(self.h, C.byref(ex), C.byref(sp), bm, nb, C.byref(st)), self.h)
        return {'exposed_ms': ex.value, 'span_ms': sp.value, 'bucket_ms': [bm[i] for i in range(nb)], 'steps': st.value}

    def step_dp(self, lr):
        """One data-parallel training step on the resident batch (bucketed RCCL all-reduce inside the library)."""
        check(self.lib.l3_step_dp(self.h, lr), self.h)

    def grad_arena(self):
        p, n = C.c_void_p(), C.c_int64()
        check(self.lib.l3_grad_arena_dev(self.h, C.byref(p), C.byref(n)), self.h)
        return p.value, n.value

    def bn_stats_pack(self):
        """(device pointer, numel) of this rank's packed BatchNorm batch means / variances (after a training forward)."""
        p, n = C.c_void_p(), C.c_int64()
        check(self.lib.l3_bn_stats_pack_dev(self.h, C.byref(p), C.byref(n)), self.h)
        return p.value, n.value

    def bn_stats_replicas(self, world):
        """Device pointer of the (world, numel) buffer the gathered statistics go to; arms the next step_update."""
        p = C.c_void_p()
        check(self.lib.l3_bn_stats_replicas_dev(self.h, int(world), C.byref(p)), self.h)
        return p.value

    def bucket_range(self, b):
        o, n = C.c_int64(), C.c_int64()
        check(self.lib.l3_bucket_range(self.h, b, C.byref(o), C.byref(n)), self.h)
        return o.value, n.value

    # -- embeddings / taps -----------------------------------------------------------------
    def embed_audio(self, audio, pool):
        a = _f32(audio)
        d = self.lib.l3_embed_dim(self.h, 0, pool[0], pool[1])
        out = np.empty((a.shape[0], d), np.float32)
        check(self.lib.l3_embed_audio(self.h, _ptr(a), a.shape[0], pool[0], pool[1], _ptr(out)), self.h)
        return out

    def embed_vision(self, video, pool=(7, 7)):
        v = _f32(video)
        d = self.lib.l3_embed_dim(self.h, 1, pool[0], pool[1])
        out = np.empty((v.shape[0], d), np.float32)
        check(self.lib.l3_embed_vision(self.h, _ptr(v), v.shape[0], pool[0], pool[1], _ptr(out)), self.h)
        return out

    def activation(self, name):
        n = C.c_int64()
        check(self.lib.l3_activation_numel(self.h, name.encode(), C.byref(n)), self.h)
        out = np.empty((n.value,), np.float32)
        check(self.lib.l3_get_activation(self.h, name.encode(), _ptr(out), n.value), self.h)
        return out

    def sync(self):
        check(self.lib.l3_sync(self.h), self.h)

    def set_tower_overlap(self, on=True):
        """Audio tower on the internal side stream beside the vision tower (default) or serialised."""
        check(self.lib.l3_set_tower_overlap(self.h, int(on)), self.h)

    def profile_enable(self, on=True):
        check(self.lib.l3_profile_enable(self.h, int(on)), self.h)

    def profile_read(self):
        out = {}
        for i, fam in enumerate(FAMILIES):
            ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
            check(self.lib.l3_profile_read(self.h, i, C.byref(ms), C.byref(n), C.byref(fl)), self.h)
            ex = C.c_double()
            check(self.lib.l3_profile_read_executed(self.h, i, C.byref(ex)), self.h)
            by = C.c_double()
            check(self.lib.l3_profile_read_bytes(self.h, i, C.byref(by)), self.h)
            out[fam] = dict(ms=ms.value, launches=n.value, flops=fl.value, executed_flops=ex.value, alg_bytes=by.value)
        return out


# -- stand-alone operators (op-level parity tests) ------------------------------------------
def op_conv2d_fwd(x, w, b, same, device=0, dtype='f32'):
    lib = load()
    x, w = _f32(x), _f32(w)
    b = _f32(b)
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    ho, wo = (h, wd) if same else (h - kh + 1, wd - kw + 1)
    y = np.empty((n, ho, wo, cout), np.float32)
    if OP_DTYPES[dtype]:
        check(lib.l3_op_conv2d_fwd_dt(device, OP_DTYPES[dtype], _ptr(x), _ptr(w), _ptr(b), _ptr(y), n, h, wd, cin, cout,
                                      kh, kw, int(same)))
    else:
        check(lib.l3_op_conv2d_fwd(device, _ptr(x), _ptr(w), _ptr(b), _ptr(y), n, h, wd, cin, cout, kh, kw, int(same)))
    return y


def op_conv2d_bwd(x, w, dy, same, device=0, dtype='f32'):
    lib = load()
    x, w, dy = _f32(x), _f32(w), _f32(dy)
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    dx, dw, db = np.empty_like(x), np.empty_like(w), np.empty((cout,), np.float32)
    if OP_DTYPES[dtype]:
        check(lib.l3_op_conv2d_bwd_dt(device, OP_DTYPES[dtype], _ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db),
                                      n, h, wd, cin, cout, kh, kw, int(same)))
    else:
        check(lib.l3_op_conv2d_bwd(device, _ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db),
                                   n, h, wd, cin, cout, kh, kw, int(same)))
    return dx, dw, db


def op_bn_relu_fwd(x, gamma, beta, relu, device=0, x_bf16=False):
    lib = load()
    x = _f32(x)
    c = x.shape[-1]
    rows = x.size // c
    y = np.empty_like(x)
    mean, var = np.empty((c,), np.float32), np.empty((c,), np.float32)
    check(lib.l3_op_bn_relu_fwd(device, _ptr(x), _ptr(_f32(gamma)), _ptr(_f32(beta)), _ptr(y), _ptr(mean),
                                _ptr(var), rows, c, int(relu), int(x_bf16)))
    return y, mean, var


def op_bn_relu_bwd(x, y, dy, gamma, mean, var, relu, device=0, beta=None, x_bf16=False):
    lib = load()
    x, y, dy = _f32(x), _f32(y), _f32(dy)
    c = x.shape[-1]
    rows = x.size // c
    dx = np.empty_like(x)
    dg, db = np.empty((c,), np.float32), np.empty((c,), np.float32)
    check(lib.l3_op_bn_relu_bwd(device, _ptr(x), _ptr(y), _ptr(dy), _ptr(_f32(gamma)), _ptr(_f32(beta)), _ptr(_f32(mean)),
                                _ptr(_f32(var)), _ptr(dx), _ptr(dg), _ptr(db), rows, c, int(relu), int(x_bf16)))
    return dx, dg, db


def op_bn_relu_pool2_fwd(x, gamma, beta, same, device=0, relu_mode=1, x_bf16=False):
    lib = load()
    x = _f32(x)
    n, h, wd, c = x.shape
    p = np.empty((n, _pool_out(h, 2, 2, same), _pool_out(wd, 2, 2, same), c), np.float32)
    mean, var = np.empty((c,), np.float32), np.empty((c,), np.float32)
    check(lib.l3_op_bn_relu_pool2_fwd(device, _ptr(x), _ptr(_f32(gamma)), _ptr(_f32(beta)), _ptr(p), _ptr(mean),
                                      _ptr(var), n, h, wd, c, int(same), int(relu_mode), int(x_bf16)))
    return p, mean, var


def op_bn_relu_pool2_bwd(x, gamma, beta, dp, same, device=0, relu_mode=1, x_bf16=False):
    lib = load()
    x, dp = _f32(x), _f32(dp)
    n, h, wd, c = x.shape
    dx = np.empty_like(x)
    dg, db, dbias = (np.empty((c,), np.float32) for _ in range(3))
    check(lib.l3_op_bn_relu_pool2_bwd(device, _ptr(x), _ptr(_f32(gamma)), _ptr(_f32(beta)), _ptr(dp), _ptr(dx),
                                      _ptr(dg), _ptr(db), _ptr(dbias), n, h, wd, c, int(same), int(relu_mode), int(x_bf16)))
    return dx, dg, db, dbias


def _pool_out(h, p, s, same):
    return -(-h // s) if same else (h - p) // s + 1


def op_maxpool_fwd(x, ph, pw, sh, sw, same, device=0):
    lib = load()
    x = _f32(x)
    n, h, wd, c = x.shape
    y = np.empty((n, _pool_out(h, ph, sh, same), _pool_out(wd, pw, sw, same), c), np.float32)
    check(lib.l3_op_maxpool_fwd(device, _ptr(x), _ptr(y), n, h, wd, c, ph, pw, sh, sw, int(same)))
    return y


def op_maxpool_bwd(x, dy, ph, pw, sh, sw, same, device=0):
    lib = load()
    x, dy = _f32(x), _f32(dy)
    n, h, wd, c = x.shape
    dx = np.empty_like(x)
    check(lib.l3_op_maxpool_bwd(device, _ptr(x), _ptr(dy), _ptr(dx), n, h, wd, c, ph, pw, sh, sw, int(same)))
    return dx


def op_bn_stats_from_partials(part, pivot, rows, eps=1e-3, device=0):
    """part: (nblk, 2, C) partial [sum, sum of squares] about pivot (C,) -> batch mean, biased variance (C,)."""
    lib = load()
    part, pivot = _f32(part), _f32(pivot)
    nblk, two, c = part.shape
    assert two == 2 and pivot.shape == (c,)
    mean, var = np.empty(c, np.float32), np.empty(c, np.float32)
    check(lib.l3_op_bn_stats_from_partials(device, _ptr(part), nblk, c, _ptr(pivot), int(rows), float(eps), _ptr(mean), _ptr(var)))
    return mean, var


def op_frontend(model_type, audio, db_max_scope='sample', device=0):
    lib = load()
    a = _f32(audio)
    n = a.shape[0]
    probe = {'cnn_L3_orig': (257, 197), 'tiny_L3': (257, 198), 'cnn_L3_kapredbinputbn': (257, 197),
             'cnn_L3_melspec1': (128, 199), 'cnn_L3_melspec2': (256, 199)}[model_type]
    out = np.empty((n, probe[0], probe[1], 1), np.float32)
    check(lib.l3_op_frontend(device, MODEL_IDS[model_type], _ptr(a), n, 0 if db_max_scope == 'sample' else 1, _ptr(out)))
    return out


def op_preprocess(video_u8=None, audio_i16=None, device=0):
    lib = load()
    v = None if video_u8 is None else np.ascontiguousarray(video_u8, dtype=np.uint8)
    a = None if audio_i16 is None else np.ascontiguousarray(audio_i16, dtype=np.int16)
    vo = None if v is None else np.empty(v.shape, np.float32)
    ao = None if a is None else np.empty(a.shape, np.float32)
    check(lib.l3_op_preprocess(device, _ptr(v), 0 if v is None else v.size, _ptr(vo),
                               _ptr(a), 0 if a is None else a.size, _ptr(ao)))
    return vo, ao
