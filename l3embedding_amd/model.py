"""Host-side mirror of l3embedding/model.py (+ the constructors of audio_model.py and
vision_model.py) on top of libl3hip.so.

Same names, argument meaning and error behaviour as the reference so that callers
(train.py:263-267, 05_generate_embedding_samples.py:153-157) can switch imports:

    MODELS[model_type](num_gpus=...) -> (model, [x_i, x_a], y)     model.py:184-195,307-313
    load_model(weights_path, model_type, src_num_gpus, tgt_num_gpus, return_io)   model.py:85-128
    load_embedding(weights_path, model_type, embedding_type, pooling_type, ...)   model.py:131-181
    convert_num_gpus(...)                                                         model.py:38-82

The model object speaks the slice of the Keras protocol those callers use: compile,
fit_generator, train_on_batch, test_on_batch, predict, get_weights/set_weights,
save_weights/load_weights, get_layer, layers, summary, to_json, name.
All arithmetic runs in the HIP library; there is no CPU compute path here.
"""
import json
from collections import OrderedDict

import os

import numpy as np

from . import _lib
from . import kerasfile
from .training_utils import get_slice_bounds

MODEL_TYPES = ('cnn_L3_orig', 'tiny_L3', 'cnn_L3_kapredbinputbn', 'cnn_L3_melspec1', 'cnn_L3_melspec2')

# audio_model.py:461-478
AUDIO_POOLING = {
    'cnn_L3_orig': {'original': (8, 8), 'short': (32, 24)},
    'cnn_L3_kapredbinputbn': {'original': (8, 8), 'short': (32, 24)},
    'cnn_L3_melspec1': {'original': (4, 8), 'short': (16, 24)},
    'cnn_L3_melspec2': {'original': (8, 8), 'short': (32, 24)},
}
VISION_POOLING = (7, 7)   # vision_model.py:212


class Input(object):
    """Placeholder for keras.layers.Input (audio_model.py:363, vision_model.py:123)."""

    def __init__(self, shape, dtype='float32', name=None):
        self.shape = (None,) + tuple(shape)
        self.dtype = dtype
        self.name = name


class Adam(object):
    """keras.optimizers.Adam as used at train.py:282 (beta_1 .9, beta_2 .999, eps 1e-8)."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=0.0):
        if (beta_1, beta_2, epsilon, decay) != (0.9, 0.999, 1e-8, 0.0):
            raise ValueError('only the keras-default Adam moments used by the reference are implemented')
        self.lr = float(lr)


class History(object):
    def __init__(self):
        self.history = {}
        self.epoch = []


class Layer(object):
    def __init__(self, name, model, tap=None):
        self.name = name
        self._model = model
        self.output = tap      # symbolic handle of the layer's output tensor


class SubModel(object):
    """'vision_model' / 'audio_model' nested models (vision_model.py:193, audio_model.py:440)."""

    def __init__(self, parent, prefix):
        self.name = prefix
        self._parent = parent

    def _names(self):
        return [n for n, _, _ in self._parent.param_table() if n.startswith(self.name + '/')]

    def get_layer(self, name):
        if not any(n.split('/')[1] == name for n in self._names()):
            raise ValueError('No such layer: ' + name)
        return Layer(name, self, tap=(self.name, name))

    def get_weights(self):
        W = self._parent._weights_dict()
        return [W[n] for n in self._names()]

    def set_weights(self, ws):
        names = self._names()
        if len(ws) != len(names):
            raise ValueError('expected %d weight arrays, got %d' % (len(names), len(ws)))
        self._parent._assign(OrderedDict(zip(names, ws)))

    def count_params(self):
        return int(sum(int(np.prod(s)) for n, s, _ in self._parent.param_table() if n.startswith(self.name + '/')))


def _host_param_table(model_type):
    """(name, shape, trainable) in the library's ledger order (layer by layer), without needing a GPU.
    Kept in lock-step with engine.hip:build_ledger (checked against the library on GPU)."""
    tab = []
    cnt = {'conv': 0, 'bn': 0}

    def conv(prefix, name, cin, cout, k):
        tab.append(('%s/%s/kernel' % (prefix, name), (k, k, cin, cout), True))
        tab.append(('%s/%s/bias' % (prefix, name), (cout,), True))

    def bn(prefix, c):
        cnt['bn'] += 1
        b = '%s/batch_normalization_%d' % (prefix, cnt['bn'])
        tab.extend([(b + '/gamma', (c,), True), (b + '/beta', (c,), True),
                    (b + '/moving_mean', (c,), False), (b + '/moving_variance', (c,), False)])

    def vgg(prefix, cin, emb):
        c = cin
        for bi, f in enumerate((64, 128, 256, 512)):
            for ci in range(2):
                if bi == 3 and ci == 1:
                    name = emb
                else:
                    cnt['conv'] += 1
                    name = 'conv2d_%d' % cnt['conv']
                conv(prefix, name, c, f, 3)
                bn(prefix, f)
                c = f

    def tiny(prefix, cin):
        c = cin
        for _ in range(3):
            cnt['conv'] += 1
            conv(prefix, 'conv2d_%d' % cnt['conv'], c, 10, 5)
            bn(prefix, 10)
            c = 10

    fe = {'cnn_L3_orig': ('spectrogram_1', 512, 0), 'tiny_L3': ('spectrogram_1', 512, 0),
          'cnn_L3_kapredbinputbn': ('spectrogram_1', 512, 0), 'cnn_L3_melspec1': ('melspectrogram_1', 2048, 128),
          'cnn_L3_melspec2': ('melspectrogram_1', 2048, 256)}[model_type]
    inputbn = model_type not in ('cnn_L3_orig', 'tiny_L3')
    if model_type == 'tiny_L3':
        tiny('vision_model', 3)
    else:
        if inputbn:
            bn('vision_model', 3)
        vgg('vision_model', 3, 'vision_embedding_layer')
    nb = fe[1] // 2 + 1
    base = 'audio_model/' + fe[0]
    tab.append((base + '/real_kernels', (fe[1], 1, 1, nb), False))
    tab.append((base + '/imag_kernels', (fe[1], 1, 1, nb), False))
    if fe[2]:
        tab.append((base + '/freq2mel', (nb, fe[2]), False))
    if model_type == 'tiny_L3':
        tiny('audio_model', 1)
        feat, head = 360 + 350, 64
    else:
        if inputbn:
            bn('audio_model', 1)
        vgg('audio_model', 1, 'audio_embedding_layer')
        feat, head = 1024, 128
    tab.extend([('dense_1/kernel', (feat, head), True), ('dense_1/bias', (head,), True),
                ('dense_2/kernel', (head, 2), True), ('dense_2/bias', (2,), True)])
    return tab


class L3Model(object):
    """An AVC model bound (lazily) to an l3_engine.  `replicas` > 1 marks the model as
    the data-parallel wrapper of training_utils.multi_gpu_model."""

    def __init__(self, model_type, seed=20180123, device=0, db_max_scope='sample', bn_zero_debias=True):
        if model_type not in MODEL_TYPES:
            raise ValueError('Invalid model type: "{}"'.format(model_type))
        self.model_type = model_type
        self.name = model_type
        self.seed = seed
        self.device = device
        self.db_max_scope = db_max_scope
        # not in the reference (fp32 only): 'bf16' = mixed precision of BASELINE configs[4] -- bf16
        # operands / fp32 accumulate in the 3x3 convolutions (include/l3hip.h, l3_config.dtype)
        self.compute_dtype = os.environ.get('L3_DTYPE', 'f32')
        # fp32 convolution algorithm (include/l3hip.h L3_FP32_CONV_*): 'f4x4' Winograd F(4x4,3x3), the default;
        # 'f2x2' F(2x2,3x3) -- ~7x lower rounding error per layer, the step ~17 % slower
        self.fp32_conv = os.environ.get('L3_FP32_CONV', 'f4x4')
        # BatchNorm moving statistics under data parallelism (include/l3hip.h L3_DP_MOVING_*): 'replicas' = one moving-average update per
        # replica and step on every rank, as multi_gpu_model's per-replica model call produces them (training_utils.py:141-157), the
        # default; 'rank_local' = one update from the rank's own shard
        self.dp_moving = os.environ.get('L3_DP_MOVING', 'replicas')
        self._inflight = None
        self.bn_zero_debias = bn_zero_debias
        self.replicas = 1
        self.optimizer = None
        self.loss = None
        self.metrics_names = ['loss']
        self.stop_training = False
        self._engine = None             # the engine holding the model state (one per fed batch size in _engines)
        self._engines = {}
        self._host_weights = None       # OrderedDict when weights were assigned before an engine exists
        self._table = None
        self.inputs = [Input((224, 224, 3), name='input_1'), Input((1, 48000), name='input_2')]
        self.outputs = [Layer('dense_2', self, tap=('head', 'dense_2'))]

    # -- engine management ---------------------------------------------------------------------
    def param_table(self):
        if self._table is None:
            self._table = _host_param_table(self.model_type)
        return self._table

    MAX_ENGINES = 3      # engines kept alive at once (one per fed batch size; ~0.4 GB of HBM per pair of batch)

    def _ensure_engine(self, batch, global_batch=0):
        """The engine for this fed batch size.  Keras keeps one set of variables whatever batch is fed, so
        when the size changes (validation_batch_size != train_batch_size, a ragged last batch) the model
        state -- weights, Adam moments and step, BatchNorm debias accumulators -- moves to the engine of
        the new size device-to-device (`l3_copy_state`) instead of being rebuilt from the weights alone."""
        key = (int(batch), int(global_batch), self.compute_dtype, self.fp32_conv)
        cur = self._engine
        if cur is not None and cur._key == key:
            return cur
        e = self._engines.get(key)
        if e is None:
            stream = None
            if self.replicas > 1:
                import torch
                if getattr(self, '_tstream', None) is None:
                    self._tstream = torch.cuda.Stream(device=self.device)
                stream = self._tstream.cuda_stream
            e = _lib.Engine(self.model_type, key[0], device=self.device, global_batch=key[1],
                            db_max_scope=self.db_max_scope, bn_zero_debias=self.bn_zero_debias, seed=self.seed,
                            stream=stream, dtype=self.compute_dtype, fp32_conv=self.fp32_conv, dp_moving=self.dp_moving)
            e._key = key
            e._trainer = None
            lib_tab = [(n, tuple(s), t) for n, s, t in e.param_table()]
            if lib_tab != [(n, tuple(s), t) for n, s, t in self.param_table()]:
                raise RuntimeError('host ledger and libl3hip ledger disagree')
            for old_key in [k for k in self._engines if self._engines[k] is not cur][:max(0, len(self._engines) + 1 - self.MAX_ENGINES)]:
                self._engines.pop(old_key).close()
            self._engines[key] = e
        if cur is not None:
            e.copy_state_from(cur)
        elif self._host_weights is not None:
            e.set_params(self._host_weights)
        self._engine = e
        self._host_weights = None
        return e

    def _drop_engines(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}
        self._engine = None

    def _weights_dict(self):
        if self._engine is not None:
            return self._engine.get_params()
        if self._host_weights is None:
            # need the library's initialisation (he_normal + kapre constants): make a batch-1 engine
            self._ensure_engine(1)
            return self._engine.get_params()
        return self._host_weights

    def _assign(self, named):
        if self._engine is not None:
            self._engine.set_params(named)
        else:
            if self._host_weights is None:
                self._host_weights = OrderedDict()
            for k, v in named.items():
                self._host_weights[k] = np.asarray(v, dtype=np.float32)
            # complete dict required before an engine exists
            missing = [n for n, _, _ in self.param_table() if n not in self._host_weights]
            if missing:
                self._ensure_engine(1)
                self._engine.set_params(named)

    # -- keras protocol ---------------------------------------------------------------------------
    @property
    def layers(self):
        vis, aud = SubModel(self, 'vision_model'), SubModel(self, 'audio_model')
        base = [Layer('input_1', self), Layer('input_2', self), vis, aud, Layer('concatenate_1', self),
                Layer('dense_1', self), Layer('dense_2', self)]
        if self.replicas > 1:
            # multi_gpu_model wrapper: [inputs, lambdas..., template model, concat] -- layers[-2] is the
            # template (model.py:77)
            return [Layer('input_1', self), Layer('input_2', self), self._template(), Layer('concatenate_1', self)]
        return base

    def _template(self):
        t = L3Model.__new__(L3Model)
        t.__dict__.update(self.__dict__)
        t.replicas = 1
        return t

    def get_layer(self, name):
        if name in ('vision_model', 'audio_model'):
            return SubModel(self, name)
        if name in ('dense_1', 'dense_2', 'concatenate_1'):
            return Layer(name, self, tap=('head', name))
        raise ValueError('No such layer: ' + name)

    def _weight_order(self):
        """[3P] keras `Model.get_weights()` order: layer by layer over the TOP-LEVEL layers, and a nested
        model's `.weights` lists all its trainable tensors before its non-trainable ones (the same order
        `save_weights` writes, kerasfile.keras_groups).  `get_layer('audio_model').get_weights()` -- a
        call on the sub-model itself -- is layer-interleaved instead (kapre's constants first:
        notebooks/extract_spectrogram_models_from_avc_models.ipynb:446); see SubModel.get_weights."""
        groups = kerasfile.keras_groups(self.param_table(), self.model_type, wrapper=self.replicas > 1)
        return [n for names in groups.values() for n in names]

    def get_weights(self):
        W = self._weights_dict()
        return [W[n] for n in self._weight_order()]

    def set_weights(self, ws):
        order = self._weight_order()
        shapes = {n: tuple(s) for n, s, _ in self.param_table()}
        if len(ws) != len(order):
            raise ValueError('You called `set_weights(weights)` on model "%s" with a weight list of length %d, '
                             'but the model was expecting %d weights.' % (self.name, len(ws), len(order)))
        named = OrderedDict()
        for n, w in zip(order, ws):
            w = np.asarray(w, dtype=np.float32)
            if tuple(w.shape) != shapes[n]:
                raise ValueError('Layer weight shape %s not compatible with provided weight shape %s (%s)' % (shapes[n], w.shape, n))
            named[n] = w
        self._assign(named)

    def count_params(self):
        return int(sum(int(np.prod(s)) for _, s, _ in self.param_table()))

    def save_weights(self, path, overwrite=True):
        """Written next to `path` and renamed onto it: a run killed mid-write leaves the previous
        `model_latest.h5` intact, and that is the file `continue_model_dir` resumes from."""
        if not overwrite and os.path.exists(path):
            raise IOError('"{}" exists and overwrite=False'.format(path))
        stem, ext = os.path.splitext(path)              # the extension picks the format (kerasfile.save_weights): keep it
        tmp = '%s.partial.%d%s' % (stem, os.getpid(), ext)
        try:
            kerasfile.save_weights(tmp, self._weights_dict(), self.param_table(), self.model_type,
                                   wrapper=self.replicas > 1)
            os.replace(tmp, path)
        except BaseException:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise

    def load_weights(self, path):
        named = kerasfile.load_weights(path, self.param_table(), self.model_type, wrapper=self.replicas > 1)
        self._assign(named)
        if self._engine is not None:
            self._engine.reset_optimizer()

    def to_json(self):
        return json.dumps({'class_name': 'Model', 'config': {'name': self.name, 'model_type': self.model_type,
                                                             'replicas': self.replicas,
                                                             'weights': [[n, list(s), t] for n, s, t in self.param_table()]},
                           'backend': 'libl3hip'})

    def get_config(self):
        return json.loads(self.to_json())['config']

    def summary(self, print_fn=print):
        tot = self.count_params()
        tr = int(sum(int(np.prod(s)) for _, s, t in self.param_table() if t))
        print_fn('Model "%s"' % self.name)
        for sub in ('vision_model', 'audio_model'):
            print_fn('  %-14s (Model)  (None, %d)   %d' % (sub, 512 if self.model_type != 'tiny_L3' else (360 if sub[0] == 'v' else 350),
                                                           SubModel(self, sub).count_params()))
        for d in ('dense_1', 'dense_2'):
            print_fn('  %-14s (Dense)  %d' % (d, sum(int(np.prod(s)) for n, s, _ in self.param_table() if n.startswith(d + '/'))))
        print_fn('Total params: {:,}'.format(tot))
        print_fn('Trainable params: {:,}'.format(tr))
        print_fn('Non-trainable params: {:,}'.format(tot - tr))

    def compile(self, optimizer, loss='categorical_crossentropy', metrics=None):
        if loss != 'categorical_crossentropy':
            raise ValueError('only categorical_crossentropy (train.py:270) is implemented')
        if isinstance(optimizer, str):
            if optimizer.lower() != 'adam':
                raise ValueError('only Adam (train.py:282) is implemented')
            optimizer = Adam()
        self.optimizer = optimizer
        self.loss = loss
        self.metrics_names = ['loss'] + (['acc'] if metrics and ('accuracy' in metrics or 'acc' in metrics) else [])

    def as_data_parallel(self, gpus):
        import os
        self.replicas = int(gpus)
        self.device = int(os.environ.get('LOCAL_RANK', self.device))     # one process per GPU
        if self._engine is not None:
            self._host_weights = self._engine.get_params()
            self._drop_engines()
        return self

    def _dist(self):
        import torch.distributed as dist
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError('a %d-replica model needs torch.distributed initialised with one process per GPU '
                               '(python -m torch.distributed.run --nproc-per-node %d ...)' % (self.replicas, self.replicas))
        if dist.get_world_size() != self.replicas:
            raise ValueError('model has %d replicas but the process group has %d ranks' % (self.replicas, dist.get_world_size()))
        return dist

    def _split(self, x, y=None):
        """[video, audio] (+ labels) -> this rank's shard (training_utils.py:121-133)."""
        v, a = x
        if self.replicas <= 1:
            return v, a, y, len(v)
        dist = self._dist()
        if getattr(x, 'global_batch', None):       # blobfeed.ShardedInputs: the feed already read only this rank's rows
            lo, hi = get_slice_bounds(x.global_batch, self.replicas, dist.get_rank())
            if len(v) != hi - lo:
                raise ValueError('rank %d expected %d rows of a %d-row batch, got %d' % (dist.get_rank(), hi - lo,
                                                                                     x.global_batch, len(v)))
            return v, a, y, x.global_batch
        lo, hi = get_slice_bounds(len(v), self.replicas, dist.get_rank())
        return v[lo:hi], a[lo:hi], (None if y is None else y[lo:hi]), len(v)

    @staticmethod
    def _is_raw(v, a):
        """uint8 frames + int16 PCM as stored in the HDF5 blobs (data/avc/sample.py:371-377)."""
        v, a = np.asarray(v), np.asarray(a)
        if v.dtype == np.uint8 and a.dtype == np.int16:
            return True
        if v.dtype == np.uint8 or a.dtype == np.int16:
            raise ValueError('raw batches need uint8 video AND int16 audio')
        return False

    def train_on_batch(self, x, y):
        self._launch_train(x, y, staged=False)
        return self._finish_train()

    def _launch_train(self, x, y, staged):
        """Enqueues one training step (returns without waiting for the device).  staged: the batch was
        already sent with Engine.stage_batch_raw() and is adopted by the step."""
        if self.optimizer is None:
            raise RuntimeError('You must compile a model before training/testing. Use `model.compile(optimizer, loss)`.')
        v, a, l, gb = self._split(x, y)
        raw = self._is_raw(v, a)
        dp = self.replicas > 1
        e = self._ensure_engine(len(v), global_batch=gb if dp else 0)
        if dp and e._trainer is None:
            # L3_DP_COMM=native (default): RCCL inside libl3hip (l3_comm_init / l3_step_dp);
            # L3_DP_COMM=torch: bucketed torch.distributed all-reduce driven from Python (the test double)
            from .training_utils import DataParallelTrainer, NativeDataParallelTrainer
            if os.environ.get('L3_DP_COMM', 'native') == 'native' and self._dist().get_backend() == 'nccl':
                e._trainer = NativeDataParallelTrainer(e, self.replicas, self._dist().get_rank())
            else:
                e._trainer = DataParallelTrainer(e, self.device, self.replicas, self._dist().get_rank(), stream=self._tstream)
        if not staged:
            if raw:      # stored dtypes straight to the GPU; train.py:186,189 scaling happens there, bit-exact
                e.upload_batch_raw(v, a, np.asarray(l).astype(np.int32))
            else:
                e.upload_batch(v, a, l)
        if dp:
            e._trainer.step(self.optimizer.lr)
        else:
            e.step_resident(self.optimizer.lr)
        self._inflight = (e, len(v), gb)

    def _defer_results(self):
        """The launched step's loss / accuracy go to a pinned slot behind it (Engine.results_enqueue): returns the handle
        `_finish_deferred` takes, or None where results cannot be deferred (engines without the entry point; the torch.distributed
        double of the exchange).  Data parallel over the library's communicator: the sums are added up over the ranks on the device,
        so the values are those of the concatenated batch (training_utils.py:165-170)."""
        e, n_local, gb = self._inflight
        if not hasattr(e, 'results_enqueue'):
            return None
        reduce = self.replicas > 1
        if reduce and not getattr(e._trainer, 'reduces_results', False):
            return None          # the torch.distributed double: the ranks' values are reduced by _finish_train
        slot = self._res_slot = 1 - getattr(self, '_res_slot', 1)
        e.results_enqueue(slot, reduce=reduce)
        return (e, slot)

    @staticmethod
    def _finish_deferred(handle):
        e, slot = handle
        return list(e.results_wait(slot))

    def _stage_next(self, x, y):
        """Sends the next (raw) batch to the device while the launched step runs.  False if it cannot be staged."""
        if self._inflight is None:
            return False
        e, n_local, _ = self._inflight
        v, a, l, _ = self._split(x, y)
        if len(v) != n_local or not self._is_raw(v, a):
            return False
        e.stage_batch_raw(v, a, np.asarray(l).astype(np.int32))
        return True

    def _finish_train(self):
        e, n_local, gb = self._inflight
        self._inflight = None
        loss, acc = e.step_results()
        if self.replicas <= 1:
            return [loss, acc]
        import torch
        # logged loss/acc are over the concatenated batch (training_utils.py:165-170)
        t = torch.tensor([loss * n_local, acc * n_local], dtype=torch.float64, device='cuda:%d' % self.device)
        self._dist().all_reduce(t)
        return [float(t[0].item()) / gb, float(t[1].item()) / gb]

    def test_on_batch(self, x, y):
        v, a, l, gb = self._split(x, y)
        e = self._ensure_engine(len(v), global_batch=gb if self.replicas > 1 else 0)
        if self._is_raw(v, a):
            e.upload_batch_raw(v, a, np.asarray(l).astype(np.int32))
            e.step_forward(training=False)
            loss, acc = e.step_results()
        else:
            loss, acc = e.eval_step(v, a, l)
        if self.replicas > 1:
            import torch
            dist = self._dist()
            t = torch.tensor([loss * len(v), acc * len(v)], dtype=torch.float64, device='cuda:%d' % self.device)
            dist.all_reduce(t)
            loss, acc = float(t[0].item()) / gb, float(t[1].item()) / gb
        return [loss, acc]

    def predict(self, x, batch_size=32, verbose=0):
        v, a = x
        n = len(v)
        e = self._ensure_engine(min(batch_size, n) if self._engine is None else self._engine.batch)
        B = e.batch
        out = np.empty((n, 2), np.float32)
        for s in range(0, n, B):
            cnt = min(B, n - s)
            vb = np.zeros((B, 224, 224, 3), np.float32)
            ab = np.zeros((B, 1, 48000), np.float32)
            vb[:cnt], ab[:cnt] = v[s:s + cnt], a[s:s + cnt]
            p, _ = e.forward(vb, ab, training=False)
            out[s:s + cnt] = p[:cnt]
        return out

    def predict_logits(self, x, training=False):
        """Pre-softmax activations of dense_2 (the quantity the 1e-3 parity bar is stated on)."""
        v, a = x
        e = self._ensure_engine(len(v))
        return e.forward(v, a, training=training)[1]

    def fit_generator(self, generator, steps_per_epoch, epochs=1, verbose=1, callbacks=None, validation_data=None,
                      validation_steps=None, initial_epoch=0, **_):
        """keras fit_generator loop as driven by train.py:408-414: per step train_on_batch, per
        epoch validation (inference-mode BN) and callbacks with logs {loss, acc, val_loss, val_acc}."""
        callbacks = list(callbacks or [])
        hist = History()
        for cb in callbacks:
            if hasattr(cb, 'set_model'):
                cb.set_model(self)
            else:
                cb.model = self
            if hasattr(cb, 'set_params'):
                cb.set_params({'epochs': epochs, 'steps': steps_per_epoch, 'verbose': verbose,
                               'do_validation': validation_data is not None, 'metrics': ['loss', 'acc', 'val_loss', 'val_acc']})
        for cb in callbacks:
            cb.on_train_begin({})
        self.stop_training = False
        max_queue_size = int(_.get('max_queue_size', 10))
        prefetchers = []
        if max_queue_size > 0:
            generator = _Prefetcher(generator, max_queue_size)
            prefetchers.append(generator)
            if validation_data is not None and hasattr(validation_data, '__next__'):
                validation_data = _Prefetcher(validation_data, max_queue_size)
                prefetchers.append(validation_data)
        try:
            return self._fit_loop(generator, steps_per_epoch, epochs, verbose, callbacks, validation_data,
                                  validation_steps, initial_epoch, hist)
        finally:
            for pf in prefetchers:
                pf.close()

    def _fit_loop(self, generator, steps_per_epoch, epochs, verbose, callbacks, validation_data, validation_steps,
                  initial_epoch, hist):
        for epoch in range(initial_epoch, epochs):
            for cb in callbacks:
                cb.on_epoch_begin(epoch, {})
            sl = sa = 0.0
            seen = 0
            pending = None      # batch fetched (and staged on the device) during the previous step
            late = None         # (step, size, handle): a step whose results are read after the NEXT step has been enqueued

            def batch_end(step, n, loss, acc):
                nonlocal sl, sa, seen
                sl += loss * n
                sa += acc * n
                seen += n
                for cb in callbacks:
                    cb.on_batch_end(step, {'batch': step, 'size': n, 'loss': loss, 'acc': acc})

            # Reading step k's results after step k + 1 has been enqueued moves the batch hooks relative to the computation:
            # begin(k) fires after step k is launched, and end(k) -- with the model already one step further -- after step k + 1 is.
            # That is invisible to hooks that only look at `logs` or the clock (the reference's own: train.py:85-131), and wrong
            # for one that sets the learning rate in on_batch_begin or saves weights in on_batch_end.  So the pipelined order is
            # taken only when every callback either leaves both batch hooks alone or declares them passive
            # (`batch_hooks_are_passive = True`, as callbacks.TimeHistory does); otherwise Keras' order holds against the
            # computation too: begin(k), step k, end(k).
            pipelined = all(_batch_hooks_passive(cb) for cb in callbacks)
            for step in range(steps_per_epoch):
                (bx, by), staged = pending if pending is not None else (next(generator)[:2], False)
                pending = None
                n = len(by)
                if not pipelined:
                    for cb in callbacks:
                        cb.on_batch_begin(step, {'batch': step, 'size': n})
                # launch, then use the device time to pull and send the next batch (never across the
                # epoch boundary: validation uploads its own batches in between)
                self._launch_train(bx, by, staged)
                handle = self._defer_results() if pipelined else None
                # the GPU now holds this step; the step before it is read (and its callbacks run) while this one computes, so the
                # device never waits for the host between steps.  Callback order is Keras': end(k - 1) before begin(k).
                if late is not None:
                    batch_end(late[0], late[1], *self._finish_deferred(late[2]))
                    late = None
                if pipelined:
                    for cb in callbacks:
                        cb.on_batch_begin(step, {'batch': step, 'size': n})
                if step + 1 < steps_per_epoch:
                    nxt = next(generator)[:2]
                    pending = (nxt, self._stage_next(*nxt))
                if handle is None:
                    batch_end(step, n, *self._finish_train())
                else:
                    late = (step, n, handle)
            if late is not None:
                self._inflight = None
                batch_end(late[0], late[1], *self._finish_deferred(late[2]))
            logs = {'loss': sl / max(seen, 1), 'acc': sa / max(seen, 1)}
            if validation_data is not None:
                vl = va = 0.0
                vseen = 0
                for _ in range(validation_steps):
                    bx, by = next(validation_data)[:2]
                    l_, a_ = self.test_on_batch(bx, by)
                    vl += l_ * len(by)
                    va += a_ * len(by)
                    vseen += len(by)
                logs['val_loss'] = vl / max(vseen, 1)
                logs['val_acc'] = va / max(vseen, 1)
            for k, v_ in logs.items():
                hist.history.setdefault(k, []).append(v_)
            hist.epoch.append(epoch)
            if verbose:
                print('Epoch %d/%d - ' % (epoch + 1, epochs) + ' - '.join('%s: %.4f' % kv for kv in sorted(logs.items())))
            for cb in callbacks:
                cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in callbacks:
            cb.on_train_end({})
        return hist


def _batch_hooks_passive(cb):
    """True if `cb` leaves on_batch_begin / on_batch_end at the base class's no-ops or declares them passive (they only read
    `logs` / the clock: nothing that depends on WHEN they run relative to the step's computation)."""
    if getattr(cb, 'batch_hooks_are_passive', False):
        return True
    from .callbacks import Callback
    for name in ('on_batch_begin', 'on_batch_end'):
        f = getattr(type(cb), name, None)
        if f is not None and f is not getattr(Callback, name):
            return False
    return True


class _Prefetcher(object):
    """[3P] keras.utils.GeneratorEnqueuer as fit_generator uses it (workers=1, max_queue_size=10):
    ONE background thread pulls batches from the Python generator into a bounded queue, so HDF5
    decoding overlaps the device step.  Order is the generator's order; an exception in the
    generator is re-raised in the consumer."""

    _END = object()

    def __init__(self, generator, max_queue_size=10):
        import queue
        import threading
        self._q = queue.Queue(maxsize=max_queue_size)
        self._stop = threading.Event()
        self._gen = generator
        self._thread = threading.Thread(target=self._run, name='l3-prefetch', daemon=True)
        self._thread.start()

    def _put(self, item):
        import queue
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self):
        try:
            for item in self._gen:
                if not self._put(item) or self._stop.is_set():
                    return
            self._put(self._END)
        except BaseException as exc:            # delivered to the consumer
            self._put(exc)

    def __iter__(self):
        return self

    def __next__(self):
        item = self._q.get()
        if item is self._END:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        return item

    def close(self):
        """Stops the thread.  The caller's generator stays usable, as with Keras (a second fit_generator on the same
        generator continues where the queue stopped pulling); whoever made the feed closes it (train.train())."""
        import queue
        self._stop.set()
        try:                                    # a producer blocked on the full queue leaves at once instead of at its next poll
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._thread.join(timeout=5.0)


class EmbeddingModel(object):
    """Output of load_embedding(): `.predict(x)` -> (N, D) embeddings (data/usc/features.py:304)."""

    def __init__(self, base, embedding_type, pool):
        self.base = base
        self.embedding_type = embedding_type
        self.pool = tuple(pool)
        self.name = 'model_embedding'

    def predict(self, x, batch_size=32, verbose=0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        e = self.base._ensure_engine(self.base._engine.batch if self.base._engine is not None else max(1, min(batch_size, len(x))))
        if self.embedding_type == 'audio':
            return e.embed_audio(x, self.pool)
        return e.embed_vision(x, self.pool)

    @property
    def output_shape(self):
        e = self.base._ensure_engine(self.base._engine.batch if self.base._engine is not None else 1)
        return (None, int(e.lib.l3_embed_dim(e.h, 1 if self.embedding_type == 'vision' else 0, self.pool[0], self.pool[1])))


# ---------------------------------------------------------------------------------------------------
# reference entry points
# ---------------------------------------------------------------------------------------------------
def multi_gpu_model(model, gpus, validate=True):
    from .training_utils import multi_gpu_model as _m
    return _m(model, gpus, validate)


def gpu_wrapper(model_f):
    """model.py:184-195"""
    def wrapped(num_gpus=0, *args, **kwargs):
        m, inp, out = model_f(*args, **kwargs)
        if num_gpus > 1:
            m = multi_gpu_model(m, gpus=num_gpus)
        return m, inp, out
    wrapped.__name__ = model_f.__name__
    return wrapped


def _construct(model_type):
    m = L3Model(model_type)
    return m, m.inputs, m.outputs[0]


@gpu_wrapper
def construct_cnn_L3_orig():
    """model.py:198-218"""
    return _construct('cnn_L3_orig')


@gpu_wrapper
def construct_cnn_L3_kapredbinputbn():
    """model.py:220-240"""
    return _construct('cnn_L3_kapredbinputbn')


@gpu_wrapper
def construct_cnn_L3_melspec1():
    """model.py:242-262"""
    return _construct('cnn_L3_melspec1')


@gpu_wrapper
def construct_cnn_L3_melspec2():
    """model.py:264-284"""
    return _construct('cnn_L3_melspec2')


@gpu_wrapper
def construct_tiny_L3():
    """model.py:286-304"""
    return _construct('tiny_L3')


MODELS = {
    'cnn_L3_orig': construct_cnn_L3_orig,
    'tiny_L3': construct_tiny_L3,
    'cnn_L3_kapredbinputbn': construct_cnn_L3_kapredbinputbn,
    'cnn_L3_melspec1': construct_cnn_L3_melspec1,
    'cnn_L3_melspec2': construct_cnn_L3_melspec2,
}


# ---------------------------------------------------------------------------------------------------
# tower-level constructors (re-exported by the reference's model.py:1-4 from vision_model.py / audio_model.py)
# ---------------------------------------------------------------------------------------------------
# (vision tower, audio tower) of every registry entry -- model.py:198-304
TOWERS = {
    'cnn_L3_orig': ('cnn_L3_orig_vision', 'cnn_L3_orig_audio'),
    'cnn_L3_kapredbinputbn': ('cnn_L3_orig_inputbn_vision', 'cnn_L3_kapredbinputbn_audio'),
    'cnn_L3_melspec1': ('cnn_L3_orig_inputbn_vision', 'cnn_L3_melspec1_audio'),
    'cnn_L3_melspec2': ('cnn_L3_orig_inputbn_vision', 'cnn_L3_melspec2_audio'),
    'tiny_L3': ('tiny_L3_vision', 'tiny_L3_audio'),
}


class TowerModel(SubModel):
    """A sub-network on its own, as `construct_cnn_L3_melspec2_audio_model()` and its siblings return it
    (audio_model.py:440-442 `Model(inputs=x_a, outputs=y_a)` named 'audio_model'; vision_model.py:193-195).  The engine
    always holds a whole AVC model, so a free-standing tower is the sub-model of a carrier registry entry that contains it;
    `L3_merge_audio_vision_models` puts two towers (and the weights set on them) back into one model."""

    def __init__(self, kind, carrier_type, prefix):
        SubModel.__init__(self, L3Model(carrier_type), prefix)
        self.kind = kind
        full = self._parent.inputs
        self.inputs = [full[0] if prefix == 'vision_model' else full[1]]
        emb = 512 if carrier_type != 'tiny_L3' else (360 if prefix == 'vision_model' else 350)
        self.output_shape = (None, emb)
        self.outputs = [Layer('flatten', self, tap=(prefix, 'output'))]

    def predict(self, x, batch_size=32, verbose=0):
        """(N, 224, 224, 3) frames or (N, 1, 48000) waveforms -> the tower's flattened output (N, 512), inference-mode
        BatchNorm: the tower's half of the concatenate input (model.py:25), read back from the engine."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = len(x)
        par = self._parent
        e = par._ensure_engine(max(1, min(batch_size, n)) if par._engine is None else par._engine.batch)
        B = e.batch
        vis = self.name == 'vision_model'
        nv = 512 if par.model_type != 'tiny_L3' else 360
        out = np.empty((n,) + self.output_shape[1:], np.float32)
        for s0 in range(0, n, B):
            cnt = min(B, n - s0)
            vb = np.zeros((B, 224, 224, 3), np.float32)
            ab = np.zeros((B, 1, 48000), np.float32)
            (vb if vis else ab)[:cnt] = x[s0:s0 + cnt]
            e.forward(vb, ab, training=False)
            h0 = e.activation('h0').reshape(B, -1)
            out[s0:s0 + cnt] = h0[:cnt, :nv] if vis else h0[:cnt, nv:]
        return out


def _tower(kind):
    carrier = [mt for mt in MODEL_TYPES if kind in TOWERS[mt]][-1]
    prefix = 'vision_model' if kind.endswith('_vision') else 'audio_model'
    m = TowerModel(kind, carrier, prefix)
    return m, m.inputs[0], m.outputs[0]


def construct_cnn_L3_orig_vision_model():
    """vision_model.py:7-99"""
    return _tower('cnn_L3_orig_vision')


def construct_cnn_L3_orig_inputbn_vision_model():
    """vision_model.py:102-195"""
    return _tower('cnn_L3_orig_inputbn_vision')


def construct_tiny_L3_vision_model():
    """vision_model.py:221-290"""
    return _tower('tiny_L3_vision')


def construct_cnn_L3_orig_audio_model():
    """audio_model.py:8-115"""
    return _tower('cnn_L3_orig_audio')


def construct_cnn_L3_kapredbinputbn_audio_model():
    """audio_model.py:118-222"""
    return _tower('cnn_L3_kapredbinputbn_audio')


def construct_cnn_L3_melspec1_audio_model():
    """audio_model.py:225-332"""
    return _tower('cnn_L3_melspec1_audio')


def construct_cnn_L3_melspec2_audio_model():
    """audio_model.py:335-442"""
    return _tower('cnn_L3_melspec2_audio')


def construct_tiny_L3_audio_model():
    """audio_model.py:490-560"""
    return _tower('tiny_L3_audio')


def L3_merge_audio_vision_models(vision_model, x_i, audio_model, x_a, model_name, layer_size=128):
    """model.py:7-35: concatenate + Dense(layer_size, relu) + Dense(2, softmax) over two towers.  The pair must be one the
    engine's ledger knows (the five registry entries of model.py:307-313, TOWERS above); weights assigned to the towers
    (`set_weights`) move into the merged model, the head keeps its he_normal initialisation."""
    pair = (getattr(vision_model, 'kind', None), getattr(audio_model, 'kind', None))
    match = [mt for mt in MODEL_TYPES if TOWERS[mt] == pair]
    if not match:
        raise ValueError('no L3 model is made of the towers {}'.format(pair))
    if layer_size != (64 if match[0] == 'tiny_L3' else 128):
        raise ValueError('layer_size={} is not the head of "{}"'.format(layer_size, match[0]))
    m = L3Model(match[0])
    m.name = model_name
    for tower in (vision_model, audio_model):
        src = tower._parent
        if src._engine is not None or src._host_weights is not None:
            W = src._weights_dict()
            m._assign(OrderedDict((n, W[n]) for n in tower._names()))
    return m, m.inputs, m.outputs[0]


def convert_num_gpus(model, inputs, outputs, model_type, src_num_gpus, tgt_num_gpus):
    """model.py:38-82"""
    if src_num_gpus <= 1 and tgt_num_gpus <= 1:
        return model, inputs, outputs
    m_new, inputs_new, output_new = MODELS[model_type]()
    m_new.set_weights(model.layers[-2].get_weights())
    if tgt_num_gpus > 1:
        m_new = multi_gpu_model(m_new, gpus=tgt_num_gpus)
    return m_new, inputs_new, output_new


def load_model(weights_path, model_type, src_num_gpus=0, tgt_num_gpus=None, return_io=False):
    """model.py:85-128"""
    if model_type not in MODELS:
        raise ValueError('Invalid model type: "{}"'.format(model_type))
    m, inputs, output = MODELS[model_type]()
    if src_num_gpus > 1:
        # only the wrapper's weight-file layout is needed here: no device check (a file written by an
        # 8-GPU run loads on any box; the reference needs src_num_gpus real GPUs even to read it)
        m = multi_gpu_model(m, gpus=src_num_gpus, validate=False)
    m.load_weights(weights_path)
    if tgt_num_gpus is not None and src_num_gpus != tgt_num_gpus:
        m, inputs, output = convert_num_gpus(m, inputs, output, model_type, src_num_gpus, tgt_num_gpus)
    if return_io:
        return m, inputs, output
    return m


def convert_audio_model_to_embedding(audio_model, x_a, model_type, pooling_type='original'):
    """audio_model.py:445-487"""
    pool_size = AUDIO_POOLING[model_type][pooling_type]
    audio_model.get_layer('audio_embedding_layer')
    m = EmbeddingModel(audio_model._parent, 'audio', pool_size)
    return m, x_a, Layer('flatten', m)


def construct_cnn_l3_orig_vision_embedding_model(vision_model, x_i):
    """vision_model.py:198-218"""
    vision_model.get_layer('vision_embedding_layer')
    m = EmbeddingModel(vision_model._parent, 'vision', VISION_POOLING)
    return m, x_i, Layer('flatten', m)


def load_embedding(weights_path, model_type, embedding_type, pooling_type, src_num_gpus=0, tgt_num_gpus=None,
                   return_io=False):
    """model.py:131-181"""
    m, inputs, output = load_model(weights_path, model_type, src_num_gpus=src_num_gpus,
                                   tgt_num_gpus=tgt_num_gpus, return_io=True)
    x_i, x_a = inputs
    if embedding_type == 'vision':
        m_embed_model = m.get_layer('vision_model')
        m_embed, x_embed, y_embed = construct_cnn_l3_orig_vision_embedding_model(m_embed_model, x_i)
    elif embedding_type == 'audio':
        m_embed_model = m.get_layer('audio_model')
        m_embed, x_embed, y_embed = convert_audio_model_to_embedding(m_embed_model, x_a, model_type, pooling_type)
    else:
        raise ValueError('Invalid embedding type: "{}"'.format(embedding_type))
    if return_io:
        return m_embed, x_embed, y_embed
    return m_embed
