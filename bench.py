#!/usr/bin/env python
"""bench.py -- AVC training pairs/sec of the MI355X-native L3-Net training step.

A "step" is one full training step (forward + backward + Adam + BN moving update, and the bucketed RCCL
gradient all-reduce when N > 1) of cnn_L3_melspec2 over one synthetic batch of 64 pairs per GPU
(BASELINE.json configs[2] at N=1, configs[3] at N=8), inputs already resident in HBM, fp32 throughout.

    python bench.py --gpus N --steps K --warmup W        # N > 1: re-executes itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W          # how the driver launches it

One process per GPU.  For N > 1 the gradient exchange is the library's own (`l3_comm_init` + `l3_step_dp`:
ncclAllReduce per gradient bucket on the communicator's HIP stream, overlapped with backward); the
ncclUniqueId travels over the launcher's env:// store, no torch process group is created
(`--comm torch` selects the torch.distributed test double instead).

Rank 0 prints ONE JSON line.  Timed region = exactly K plain steps between barrier + torch.cuda.synchronize();
the per-kernel numbers come from `--roofline-steps` FURTHER steps with hipEvents around every launch and the
two towers serialised on one stream (which is what a per-kernel duration means and what
`rocprofv3 --kernel-trace --stats -- python bench.py --serial` shows, profiles/*_kernel_stats.csv):
  roofline       the top rocprof symbol of the step.  frac = ISSUED flops / (duration * MFMA peak);
                 `algorithmic_frac` uses SURVEY 8(d)'s direct-convolution flop count instead (they differ only
                 for Winograd: forward / data gradient run F(4x4,3x3), which issues 9/36 of the direct multiplies plus
                 tile padding; the weight gradient F(3x3,2x2), 16/36).  `traffic` is HBM bytes per launch from the
                 builder's committed rocprofv3 PMC pass (profiles/pmc_traffic.json, `traffic_source` says so) -- PMC
                 counters cannot be collected from inside this process
  kernels        the same two fractions for each convolution family, with launches and ms per step
  cpu_baseline   SURVEY 8(d) protocol: the identical fp32 step in PyTorch-CPU/oneDNN at batch 64 on the host
                 cores (thread count chosen by a short sweep up to all cores), >= 3 timed steps
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
# before anything initialises the HIP runtime: enough hardware queues for the engine's streams (two towers, RCCL,
# staging) not to share one -- shared queues turn one stream's event waits into another's stalls (csrc/comm.hip)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

F_TRAIN_GFLOP_PER_PAIR = 124.5519      # 3 x (convs + head) + DFT + mel
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0         # dense bf16 MFMA (same table)
PEAK_HBM_TBS = 8.0                     # HBM3E spec (same table)
MEASURED_HBM_TBS = 6.29                # ... and what a streaming kernel reaches on it (MI355X_MICROARCH.md, HBM section)
PROTOCOL_MIN_WARMUP, PROTOCOL_MIN_STEPS = 10, 50


def synthetic_raw(batch, seed, rank):
    """int16 PCM U{-32768..32767}, uint8 frames U{0..255}, Bernoulli(0.5) labels."""
    rng = np.random.RandomState(seed + rank)
    pcm = rng.randint(-32768, 32768, size=(batch, 1, 48000)).astype(np.int16)
    frm = rng.randint(0, 256, size=(batch, 224, 224, 3)).astype(np.uint8)
    lab = rng.randint(0, 2, size=(batch,))
    labels = np.stack([lab, 1 - lab], axis=1).astype(np.int32)
    return frm, pcm, labels


HEAD_SCALE = 1.0 / 64      # see live_head()


def live_head(eng, scale=HEAD_SCALE):
    """SURVEY 8(d)'s he_normal head on 8(d)'s U[-1,1) inputs puts EVERY sample of the synthetic batch outside the 1e-7
    probability clip of categorical_crossentropy (train.py:269-284): dlogits == 0, the loss never moves, and every backward kernel
    multiplies zeros.  The reference's real runs do not sit there.  The one deviation from 8(d): `dense_2/kernel`
    starts at `scale` x its he_normal draw, so the initial logits are small, every sample has a live loss gradient and the
    loss falls from the first step.  Everything else (inputs, seeds, all other weights) is 8(d)'s."""
    w = eng.get_param('dense_2/kernel', (128, 2))
    eng.set_param('dense_2/kernel', w * np.float32(scale))


def dlogits_nonzero_frac(probs, labels_onehot, eps=1e-7):
    """Share of the samples whose loss gradient is not clipped away: eps < p(true class) < 1 - eps (train.py:282-284: keras'
    categorical_crossentropy clips the probabilities to [1e-7, 1 - 1e-7]; outside, d loss / d logits is exactly zero)."""
    p = (np.asarray(probs, np.float64) * np.asarray(labels_onehot, np.float64)).sum(axis=1)
    return float(np.mean((p > eps) & (p < 1.0 - eps)))


def _time_cpu_steps(tr, v, a, l, n_timed, warm=1):
    for _ in range(warm):
        tr.step(v, a, l, 1e-4)
    times = []
    for _ in range(n_timed):
        t0 = time.perf_counter()
        tr.step(v, a, l, 1e-4)
        times.append(time.perf_counter() - t0)
    return times


def cpu_baseline(model_type, batch=64, quick=False):
    """SURVEY 8(d) 'CPU baseline (timed beside it)'.  The reference's Keras-2.0.9/TF-1.4 CPU path cannot be
    installed, so stand-in (ii) is timed: the identical fp32 training step in PyTorch-CPU /
    oneDNN (`oracle/torch_cpu.py`, a stronger baseline than TF-1.4's Eigen kernels).  Protocol: same
    synthetic inputs, batch 64 (and batch 16 for cnn_L3_orig = BASELINE configs[0]), 1 warm-up + 3 timed
    steps, pairs/s = B / median step seconds; thread count = the best of a one-step sweep at batch 16 over
    {16, 32, 64, 128, all hardware threads}, stopped once a step is 2.5x slower than the best so far (oneDNN on
    a 2-socket host gets slower, not faster, beyond ~16-32 threads at these batch sizes)."""
    import torch
    from oracle import l3_oracle as o
    from oracle.torch_cpu import TorchCpuTrainer
    host = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    P = o.init_params(model_type, seed=20180123)
    v, a, l = o.synthetic_batch(batch)
    tr = TorchCpuTrainer(model_type, P)
    sweep = {}
    sb = min(16, batch)
    for nt in sorted(set(t for t in (16, 32, 64, 128, host) if t <= host)):
        torch.set_num_threads(nt)
        sweep[nt] = _time_cpu_steps(tr, v[:sb], a[:sb], l[:sb], 1, warm=1 if not sweep else 0)[0]
        if sweep[nt] > 2.5 * min(sweep.values()):     # past the knee more threads only get slower (measured on the
            break                                      # 256-thread box: 3.5 / 3.9 / 5.7 / 10.4 / 95 s per step)
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = _time_cpu_steps(tr, v, a, l, 1 if quick else 3, warm=0 if quick else 1)
    med = float(np.median(times))
    quota = host_cpu_info().get("cgroup_cpu_quota_cores")
    on = "%d threads on a %g-core cgroup quota (%d hardware threads visible)" % (best, quota, host) if quota else "%d of %d host threads" % (best, host)
    out = {"value": batch / med, "unit": "pairs/s", "cores": best if not quota else min(float(best), float(quota)), "threads": best,
           "quota_cores": quota, "host_cores": host, "kind": "port",
           "batch": batch, "timed_steps": len(times), "median_s_per_step": round(med, 3),
           "thread_sweep_s_per_step_at_batch_%d" % sb: {str(k): round(t, 3) for k, t in sweep.items()},
           "sample": "CPU reference stand-in (Keras path not runnable: see SURVEY 8(c)): %d timed fp32 training steps "
                     "(fwd+bwd+Adam) of %s at batch %d in PyTorch-CPU/oneDNN (oracle/torch_cpu.py) on %s, "
                     "median %.2f s/step" % (len(times), model_type, batch, on, med)}
    if not quick:       # BASELINE configs[0]: cnn_L3_orig at batch 16
        P0 = o.init_params('cnn_L3_orig', seed=20180123)
        tr0 = TorchCpuTrainer('cnn_L3_orig', P0)
        t0 = _time_cpu_steps(tr0, v[:16], a[:16], l[:16], 3, warm=1)
        out["cnn_L3_orig_batch16"] = {"value": 16 / float(np.median(t0)), "unit": "pairs/s", "cores": best,
                                      "median_s_per_step": round(float(np.median(t0)), 3), "timed_steps": 3}
    torch.set_num_threads(default_threads)
    return out


def respawn_under_launcher(n):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py <same args>`."""
    from l3embedding_amd import launch
    launch.respawn_under_launcher(n, [os.path.abspath(__file__)], sys.argv[1:])


def host_cpu_info():
    """What the host offers the CPU leg: hardware threads, the affinity mask, and the cgroup CPU quota (a container
    limited to fewer cores than it can see makes a thread sweep get *slower* with more threads)."""
    info = {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None,
            "cgroup_cpu_quota_cores": None}
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:                  # cgroup v2: "<quota|max> <period>"
            q, per = fh.read().split()
            info["cgroup_cpu_quota_cores"] = None if q == 'max' else float(q) / float(per)
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            info["cgroup_cpu_quota_cores"] = None if q <= 0 else q / per
        except Exception:
            pass
    return info


def _agree(store, tag, rank, world, ok):
    """True on every rank iff `ok` was true on every rank (a flag per rank in the launcher's env:// store)."""
    run = os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')
    store.set('l3hip/bench/%s/%s/%d' % (run, tag, rank), b'1' if ok else b'0')
    return all(bytes(store.get('l3hip/bench/%s/%s/%d' % (run, tag, r))) == b'1' for r in range(world))


class Ranks(object):
    """barrier / max-over-ranks for the timed region, over whichever communicator the run uses."""

    def __init__(self, args, eng, world, rank, local_rank, tstream, native_factory=None, torch_factory=None,
                 torch_backend='nccl'):
        import torch
        self.world, self.rank, self.torch, self.dist, self.native = world, rank, torch, None, None
        self.eng = eng if hasattr(eng, 'sync') else None
        self.fallback = None                    # why the torch.distributed double runs instead of the in-library exchange
        from l3embedding_amd import training_utils
        native_factory = native_factory or training_utils.NativeDataParallelTrainer
        torch_factory = torch_factory or (lambda: training_utils.DataParallelTrainer(eng, local_rank, world, rank, stream=tstream))

        def torch_group():
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            kw = {'device_id': torch.device('cuda', local_rank)} if torch_backend == 'nccl' else {}
            dist.init_process_group(backend=torch_backend, rank=rank, world_size=world, **kw)
            self.dist = dist

        if world > 1 and args.comm == 'torch':
            torch_group()
        if (world > 1 and args.comm == 'native') or args.force_comm:
            # Every rank must end up on the SAME exchange: a rank that fell back alone would sit in a different collective
            # from the others and the job would hang.  Two agreements over the launcher's store: (1) before anybody enters
            # ncclCommInitRank -- which blocks until all ranks arrive -- that every rank can load librccl at all; (2) after
            # it, that every rank's communicator came up.  Any "no" sends ALL ranks to the torch.distributed double, and the
            # JSON line says so (`comm.fallback`).
            store = training_utils._env_store(rank, world) if world > 1 else None
            agree = (lambda tag, ok: _agree(store, tag, rank, world, ok)) if store is not None else (lambda tag, ok: ok)
            err = None
            try:
                from l3embedding_amd import _lib
                _lib.comm_unique_id()           # dlopens librccl in this process (the id itself is discarded)
            except Exception as exc:
                err = 'librccl not loadable: %s' % exc
            if agree('preflight', err is None):
                try:
                    self.native = native_factory(eng, world, rank)
                except Exception as exc:
                    err = 'communicator: %s' % exc
                if not agree('init', self.native is not None):
                    if self.native is not None:
                        try:
                            eng.comm_destroy()
                        except Exception:
                            pass
                    self.native = None
            if self.native is not None:
                self.trainer = self.native
                return
            self.fallback = err or 'another rank could not bring up the in-library communicator'
            sys.stderr.write('bench.py rank %d: in-library RCCL exchange unavailable (%s); every rank uses torch.distributed\n'
                             % (rank, self.fallback))
            if world > 1:
                torch_group()
        self.trainer = torch_factory()

    def barrier(self):
        if self.eng is not None:
            self.eng.sync()          # the library's host wait (sleep-poll: the rank's thread costs no CPU while the GPU works)
        if self.native is not None and self.world > 1:
            self.native.barrier()
        elif self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max(self, x):
        if self.world == 1:
            return x
        if self.native is not None:
            return self.native.allreduce([x], 'max')[0]
        t = self.torch.tensor([x], dtype=self.torch.float64, device='cuda')
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def comm_desc(self, eng):
        if self.native is not None:
            info = eng.comm_info()
            from l3embedding_amd import _lib
            return {"backend": "libl3hip l3_comm_* (RCCL, in-library bucketed all-reduce)", "ranks": info['world'],
                    "librccl": info['library'], "rccl_version": _lib.load().l3_comm_version(),
                    # what bounds RCCL's CU footprint beside the persistent convolution grids; None = RCCL's default
                    "rccl_env": {k: os.environ.get(k) for k in ('NCCL_MAX_NCHANNELS', 'NCCL_MIN_NCHANNELS', 'NCCL_NTHREADS')},
                    "dp_moving": getattr(eng, 'dp_moving', None)}
        if self.dist is not None:
            return {"backend": "torch.distributed nccl (RCCL), Python-driven buckets", "ranks": self.dist.get_world_size(),
                    "fallback": self.fallback}
        return {"backend": "none (single GPU)", "ranks": 1}

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def tower_bench(args, eng, B, world, rank, ranks, json_out):
    """One sub-network alone (replicas only: nothing is exchanged between ranks)."""
    tower = 'audio' if args.workload == 'audio_tower' else 'vision'
    peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == 'f32' else PEAK_BF16_MFMA_TFLOPS

    def run(backward, steps):
        ranks.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.tower_step(tower, backward)
        eng.sync()
        ranks.barrier()
        return time.perf_counter() - t0

    run(True, args.warmup)
    t_fwd = ranks.max(run(False, args.steps))
    elapsed = ranks.max(run(True, args.steps))
    eng.set_tower_overlap(False)
    eng.profile_enable(True)
    run(True, args.roofline_steps)
    prof = eng.profile_read()
    if rank == 0:
        ig_ms = prof['conv_fwd']['ms'] + prof['conv_dgrad']['ms']
        ex = prof['conv_fwd']['executed_flops'] + prof['conv_dgrad']['executed_flops']
        al = prof['conv_fwd']['flops'] + prof['conv_dgrad']['flops']
        json_out.write(json.dumps({
            "metric": "%s-tower samples/sec, training-mode forward + backward (stand-in loss = mean of the tower output, "
                      "no optimizer step)" % tower,
            "value": B * world * args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "cnn_L3_melspec2 %s tower only (%s), batch %d per GPU, inputs resident in HBM" %
                                   (tower, "mel front-end + audio conv kernels" if tower == 'audio' else "vision conv kernels", B),
                       "parallelism": "replicas%d" % world},
            "forward_only": {"value": B * world * args.steps / t_fwd, "unit": "samples/s", "ms_per_step": 1e3 * t_fwd / args.steps},
            "roofline": {"bound": "mfma", "achieved": ex / (ig_ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                         "frac": ex / (ig_ms * 1e-3) / 1e12 / peak, "algorithmic_frac": al / (ig_ms * 1e-3) / 1e12 / peak,
                         "traffic": None, "kernel": "forward + dgrad convolution launches of the tower",
                         "measured": "%d further fwd+bwd passes with hipEvents around every launch" % args.roofline_steps},
            "kernel_ms_per_step": {k: v['ms'] / args.roofline_steps for k, v in prof.items()},
            "cpu_baseline": None}) + '\n')
        json_out.flush()
    ranks.close()
    eng.close()


def secondary_lines(args, local_rank, tstream, steps=20, warmup=5, prof_steps=3):
    """the other single-GPU configurations of BASELINE.json, timed after the headline's region on engines of
    their own (N = 1 only; the headline is untouched): configs[1] = cnn_L3_melspec2 audio tower only, batch 64, fp32
    (`l3_tower_step`, forward + backward from mean(output), no optimizer step) and the per-GPU shard of configs[4] = the full
    step at 128 pairs in bf16 mixed precision (live head like the headline)."""
    from l3embedding_amd import _lib
    out = {}

    try:
        pmc = json.load(open(os.path.join(HERE, 'profiles', 'pmc_traffic.json')))
    except Exception:
        pmc = {}

    def hbm(prof, fams, key):       # TB/s of a kernel family: algorithmic bytes (engine ledger) and counter bytes (committed PMC pass) / time
        ms = sum(prof[f]['ms'] for f in fams)
        n = sum(prof[f]['launches'] for f in fams)
        ab = sum(prof[f].get('alg_bytes', 0.0) for f in fams)
        tr = pmc.get(key, {}).get('hbm_bytes_per_launch') if isinstance(pmc.get(key), dict) else None
        avg_s = ms * 1e-3 / n if n else None
        return {"alg_bytes_per_launch": ab / n if n else None, "alg_hbm_tbs": ab / (ms * 1e-3) / 1e12 if ms > 0 else None,
                "traffic": tr, "hbm_tbs": None if tr is None or not avg_s else tr / avg_s / 1e12,
                "hbm_frac": None if tr is None or not avg_s else tr / avg_s / 1e12 / PEAK_HBM_TBS,
                "traffic_over_algorithmic": None if tr is None or not ab else tr / (ab / n)}

    def conv_frac(prof, peak, key=None):
        ms = prof['conv_fwd']['ms'] + prof['conv_dgrad']['ms']
        ex = prof['conv_fwd']['executed_flops'] + prof['conv_dgrad']['executed_flops']
        al = prof['conv_fwd']['flops'] + prof['conv_dgrad']['flops']
        return dict({"kernel": "forward + data-gradient convolution launches", "frac": ex / (ms * 1e-3) / 1e12 / peak,
                     "algorithmic_frac": al / (ms * 1e-3) / 1e12 / peak, "ms_per_step": ms / prof_steps, "peak": peak, "unit": "TFLOP/s"},
                    **hbm(prof, ['conv_fwd', 'conv_dgrad'], key))

    def run(eng, step, n):
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        eng.sync()
        return time.perf_counter() - t0

    def profile(eng, step):
        eng.set_tower_overlap(False)
        eng.profile_enable(True)
        run(eng, step, prof_steps)
        prof = eng.profile_read()
        eng.profile_enable(False)
        return prof

    # ---- configs[1]: audio tower only, batch 64, fp32 ----
    B = 64
    eng = _lib.Engine(args.model, B, device=local_rank, seed=20180123, stream=tstream.cuda_stream, dtype='f32', fp32_conv=args.fp32_conv)
    frm, pcm, lab = synthetic_raw(B, 20180123, 0)
    eng.upload_batch_raw(frm, pcm, lab)
    step = lambda: eng.tower_step('audio', True)
    run(eng, step, warmup)
    dt = run(eng, step, steps)
    dt_fwd = run(eng, lambda: eng.tower_step('audio', False), steps)
    prof = profile(eng, step)
    out["configs[1] audio tower b64 fp32"] = {
        "metric": "audio-tower samples/sec, training-mode forward + backward (stand-in loss = mean of the tower output, no optimizer step)",
        "value": B * steps / dt, "unit": "samples/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup, "dtype": "f32",
        "forward_only": {"value": B * steps / dt_fwd, "ms_per_step": 1e3 * dt_fwd / steps},
        "roofline": conv_frac(prof, PEAK_FP32_MFMA_TFLOPS, 'conv_wino4')}
    eng.close()
    # ---- configs[4]'s per-GPU shard: full step, 128 pairs, bf16 mixed precision ----
    B = 128
    eng = _lib.Engine(args.model, B, device=local_rank, global_batch=B, seed=20180123, stream=tstream.cuda_stream, dtype='bf16')
    frm, pcm, lab = synthetic_raw(B, 20180123, 0)
    eng.upload_batch_raw(frm, pcm, lab)
    if args.head_scale != 1.0:
        live_head(eng, args.head_scale)
    step = lambda: eng.step_resident(args.lr)
    run(eng, step, warmup)
    l0 = eng.step_results()[0]
    dt = run(eng, step, steps)
    l1, _, probs, _ = eng.step_results(want_probs=True)
    prof = profile(eng, step)
    wg_ms = prof['conv_wgrad']['ms']
    out["configs[4] shard b128 bf16"] = {
        "metric": "AVC training pairs/sec, full step, bf16 conv operands / f32 accumulate (one GPU's 128-pair shard of configs[4])",
        "value": B * steps / dt, "unit": "pairs/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
        "dtype": "bf16 conv operands / f32 accumulate (everything else f32)",
        "loss_before": l0, "loss_after": l1, "dlogits_nonzero_frac": dlogits_nonzero_frac(probs, lab),
        "roofline": conv_frac(prof, PEAK_BF16_MFMA_TFLOPS, 'conv_bf16'),
        "conv_wgrad": dict({"frac": prof['conv_wgrad']['executed_flops'] / (wg_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                            "ms_per_step": wg_ms / prof_steps}, **hbm(prof, ['conv_wgrad'], 'conv_wgrad9t_bf16')),
        "elementwise": {"bound": "hbm", "ms_per_step": prof['elementwise']['ms'] / prof_steps,
                        "alg_bytes_per_step": prof['elementwise'].get('alg_bytes', 0.0) / prof_steps,
                        "achieved": prof['elementwise'].get('alg_bytes', 0.0) / (prof['elementwise']['ms'] * 1e-3) / 1e12,
                        "peak": PEAK_HBM_TBS, "unit": "TB/s"},
        "kernel_ms_per_step": {k: v['ms'] / prof_steps for k, v in prof.items()}}
    eng.close()
    return out


def claim_stdout():
    """The contract is ONE JSON line on stdout, and libraries write there too (RCCL prints a version banner
    through C stdio when its first communicator comes up, which lands after anything Python printed).  Keep a
    private handle on the real stdout for the JSON line and point fd 1 at stderr for everything else."""
    sys.stdout.flush()
    keep = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    return keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch-per-gpu', type=int, default=64)
    ap.add_argument('--model', default='cnn_L3_melspec2')
    ap.add_argument('--lr', type=float, default=1e-4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--quick-cpu-baseline', action='store_true', help='one timed CPU step instead of the 8(d) protocol')
    ap.add_argument('--workload', default='full', choices=['full', 'audio_tower', 'vision_tower'],
                    help="full: the AVC training step (the metric).  audio_tower / vision_tower: one sub-network alone, "
                         "training-mode forward + backward from mean(output), no optimizer step -- SURVEY 8(d) config 2 "
                         "(BASELINE configs[1]); reported as its own line, never as the AVC metric")
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help="f32: the headline configuration (BASELINE.json configs[2]/[3]); bf16: mixed precision of "
                         "configs[4] (bf16 conv operands, fp32 accumulate; use --batch-per-gpu 128) -- its own line, never "
                         "the fp32 metric")
    ap.add_argument('--fp32-conv', default='f4x4', choices=['f4x4', 'f2x2', 'f2x2_bf16x6'],
                    help="l3_config.fp32_conv: Winograd F(4x4,3x3) (default, the product configuration) or F(2x2,3x3) (lower rounding "
                         "error, slower) for forward / data gradient of the 14 3x3 layers -- its own line, not the headline")
    ap.add_argument('--head-scale', type=float, default=HEAD_SCALE,
                    help="dense_2/kernel starts at this multiple of its he_normal draw so that the synthetic batch has live loss "
                         "gradients (see live_head(); 1.0 = SURVEY 8(d)'s untouched head, where every sample is clipped and backward "
                         "runs on zeros)")
    ap.add_argument('--no-saturated', action='store_true', help='skip the second timed region on the untouched (saturated) head')
    ap.add_argument('--no-secondary', action='store_true', help='skip the configs[1] / configs[4]-shard lines under "secondary"')
    ap.add_argument('--roofline-steps', type=int, default=5,
                    help='further steps, outside the timed region, with per-launch hipEvents and the towers serialised')
    ap.add_argument('--serial', action='store_true', help='run the timed region with the towers serialised too')
    ap.add_argument('--comm', default='native', choices=['native', 'torch'],
                    help="N > 1: native = RCCL inside libl3hip (l3_comm_init / l3_step_dp); torch = torch.distributed double")
    ap.add_argument('--force-comm', action='store_true', help='N = 1 through the data-parallel path (world-1 communicator)')
    ap.add_argument('--watchdog-seconds', type=int, default=1500,
                    help='a rank that is still running after this long dumps every thread\'s stack to stderr and exits: a collective '
                         'that never completes becomes a failed run with a traceback, not a hang (0 = off)')
    args = ap.parse_args()
    if args.watchdog_seconds > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog_seconds, exit=True)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world == 1 and args.gpus > 1:
        respawn_under_launcher(args.gpus)
    if world != args.gpus:
        raise SystemExit('--gpus %d but the launcher started %d ranks' % (args.gpus, world))

    json_out = claim_stdout()
    import torch
    from l3embedding_amd import _lib
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an AMD GPU (no CPU fallback)')
    torch.cuda.set_device(local_rank)

    B = args.batch_per_gpu
    # a real side stream shared by the engine's kernels and (torch double) the collectives' dependencies
    tstream = torch.cuda.Stream(device=local_rank)
    assert tstream.cuda_stream != 0
    eng = _lib.Engine(args.model, B, device=local_rank, global_batch=B * world, seed=20180123,
                      stream=tstream.cuda_stream, dtype=args.dtype, fp32_conv=args.fp32_conv)
    frm, pcm, lab = synthetic_raw(B, 20180123, rank)
    eng.upload_batch_raw(frm, pcm, lab)          # uint8/int16 -> fp32 on the GPU (train.py:186,189)
    ranks = Ranks(args, eng, world, rank, local_rank, tstream)
    trainer = ranks.trainer
    if args.workload != 'full':
        return tower_bench(args, eng, B, world, rank, ranks, json_out)

    if args.serial:
        eng.set_tower_overlap(False)
    init_params = eng.get_params() if args.head_scale != 1.0 else None      # 8(d)'s untouched initialisation (saturated regime below)
    if args.head_scale != 1.0:
        live_head(eng, args.head_scale)

    def timed_region(record_losses):
        """W untimed steps, then EXACTLY K plain steps between barrier + torch.cuda.synchronize().  record_losses: every step's
        loss is copied to a pinned slot behind it and read one step late (l3_step_results_enqueue / _wait, what fit_generator does
        for train.py:408-414's per-batch logs) -- no host wait on the step that is running."""
        for _ in range(args.warmup):
            trainer.step(args.lr)
        ranks.barrier()
        _, _, p0, _ = eng.step_results(want_probs=True)
        losses = []
        ranks.barrier()
        t0, c0 = time.perf_counter(), time.process_time()
        for k in range(args.steps):
            trainer.step(args.lr)
            if record_losses:
                eng.results_enqueue(k & 1)
                if k:
                    losses.append(eng.results_wait((k - 1) & 1)[0])
        ranks.barrier()
        cores = (time.process_time() - c0) / max(time.perf_counter() - t0, 1e-9)     # this rank's process, all its threads
        dt = ranks.max(time.perf_counter() - t0)
        if record_losses:
            losses.append(eng.results_wait((args.steps - 1) & 1)[0])
        loss_, _, p1, _ = eng.step_results(want_probs=True)
        return dt, cores, losses, loss_, dlogits_nonzero_frac(p0, lab), dlogits_nonzero_frac(p1, lab)

    # ---- timed region: exactly K plain steps (no profiling events) ---------------------------------------------
    elapsed, host_cores, losses, loss, live0, live1 = timed_region(True)
    # ---- per-kernel durations: further steps, hipEvents around every launch, towers serialised ---------------
    prof, prof_steps = None, args.roofline_steps
    if prof_steps > 0:
        eng.set_tower_overlap(False)
        eng.profile_enable(True)
        for _ in range(prof_steps):
            trainer.step(args.lr)
        ranks.barrier()
        eng.sync()
        prof = eng.profile_read()
        eng.profile_enable(False)
        eng.set_tower_overlap(not args.serial)
    # ---- the exchange, as the step sees it (in-library communicator only): further steps in l3_comm_timing mode ----
    comm_timing = None
    if ranks.native is not None:
        eng.comm_timing(True)
        for _ in range(max(prof_steps, 3)):
            trainer.step(args.lr)
        comm_timing = eng.comm_timing_read()
        eng.comm_timing(False)
        ranks.barrier()
    # ---- the round-1..4 regime for comparison: 8(d)'s untouched he_normal head, every sample outside the clip, dlogits == 0 ----
    saturated = None
    if init_params is not None and not args.no_saturated:
        eng.set_params(init_params)
        eng.reset_optimizer()
        s_elapsed, _, _, s_loss, s_live0, s_live1 = timed_region(False)
        saturated = {"value": B * world * args.steps / s_elapsed, "ms_per_step": 1e3 * s_elapsed / args.steps, "final_loss": s_loss,
                     "dlogits_nonzero_frac": s_live1}

    if rank == 0:
        peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == 'f32' else PEAK_BF16_MFMA_TFLOPS
        value = B * world * args.steps / elapsed
        out = {
            "metric": "AVC training pairs/sec (1s audio + 224x224 frame)",
            # which regime `value` is: rounds 1-4 timed SURVEY 8(d)'s untouched head, where the loss gradient of the synthetic
            # batch is clipped to zero and backward multiplies zeros; since round 5 `value` is the live-gradient regime and the old
            # series continues under `value_saturated_head` -- compare like with like across rounds
            "value_regime": "live loss gradients (head_scale %g); rounds <= 4 compare with value_saturated_head" % args.head_scale,
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            # discard >= 10 warm-up steps, time >= 50.  The caller's --steps/--warmup are honoured as given;
            # a shorter run says so here
            "protocol_ok": bool(args.warmup >= PROTOCOL_MIN_WARMUP and args.steps >= PROTOCOL_MIN_STEPS),
            "dtype": "f32" if args.dtype == 'f32' else "bf16 conv operands / f32 accumulate (everything else f32)",
            "data": "synthetic",
            # host CPU this rank's process took during the timed region, in cores (the library's host waits sleep between
            # hipStreamQuery calls -- csrc/knobs.h stream_wait; what is left is a runtime thread of ROCr)
            "host_cpu_cores_per_rank": round(host_cores, 2),
            "config": {"workload": "full %s AVC training step (audio+vision+fusion, fwd+bwd+Adam), batch %d per GPU, "
                                   "global batch %d, %s, inputs resident in HBM" %
                                   (args.model, B, B * world, "fp32" if args.dtype == 'f32' else "bf16 mixed precision"),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "fp32_conv": args.fp32_conv},
            "comm": dict(ranks.comm_desc(eng), **({} if comm_timing is None else {
                # hipEvents of `steps` further steps (each waited for): how long Adam waited for the wire after backward was done,
                # first collective start -> last collective done, and each bucket's all-reduce (head, vision 4..1, audio 4..1)
                "exposed_ms": comm_timing['exposed_ms'], "span_ms": comm_timing['span_ms'],
                "bucket_allreduce_ms": [round(x, 4) for x in comm_timing['bucket_ms']], "timing_steps": comm_timing['steps']})),
            "tower_overlap": not args.serial,
            "step_fraction_of_mfma_peak_algorithmic": value / world * F_TRAIN_GFLOP_PER_PAIR * 1e9 / (peak * 1e12),
            "final_loss": loss,
            # live loss gradients: the loss of every timed step, read one step late
            "loss_first": losses[0] if losses else None, "loss_last": losses[-1] if losses else None,
            "loss_strictly_decreasing": bool(losses and all(b < a for a, b in zip(losses, losses[1:]))),
            "dlogits_nonzero_frac": min(live0, live1),
            "dlogits_nonzero_frac_before_after": [live0, live1],
            "head_scale": args.head_scale,
        }
        out["config"]["deviation_from_8d"] = (None if args.head_scale == 1.0 else
            "dense_2/kernel starts at %g x its he_normal draw (live loss gradients; with the untouched head every sample of the "
            "synthetic batch is outside the 1e-7 clip and backward multiplies zeros)" % args.head_scale)
        if saturated is not None:
            out["value_saturated_head"] = saturated["value"]
            out["saturated_head"] = saturated
        if prof is not None:
            traffic = {}
            tpath = os.path.join(HERE, 'profiles', 'pmc_traffic.json')     # committed PMC pass (scripts/pmc_conv.sh)
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath))
                except Exception:
                    traffic = {}

            alu = {}
            apath = os.path.join(HERE, 'profiles', 'pmc_alu.json')          # committed PMC pass (scripts/pmc_summarize.py)
            if os.path.exists(apath):
                try:
                    alu = json.load(open(apath))
                except Exception:
                    alu = {}

            from l3embedding_amd import _build

            def stale(doc, srcs):       # True: the PMC summary was taken on another version of the kernel's source (or carries no key)
                st = doc.get('stamp') if isinstance(doc, dict) else None
                have = st.get('kernel_files_sha16', {}) if isinstance(st, dict) else {}
                return not all(have.get(f) == _build.source_hash([f]) for f in srcs)

            SRC_OF = {'conv_wino4': ['conv_wino4.hip'], 'conv_wgrad_wino': ['conv_wgrad_wino.hip'], 'conv_bf16': ['conv_bf16_halo.hip'],
                      'conv_wgrad9t_bf16': ['conv_wgrad_bf16.hip']}

            def family(names, kernel, traffic_key, peak=peak):
                ms = sum(prof[n]['ms'] for n in names)
                n = sum(prof[n]['launches'] for n in names)
                al = sum(prof[n]['flops'] for n in names)
                ex = sum(prof[n]['executed_flops'] for n in names)
                tf = lambda fl: fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                tr = traffic.get(traffic_key, {}).get('hbm_bytes_per_launch') if isinstance(traffic.get(traffic_key), dict) else None
                ab = sum(prof[n_].get('alg_bytes', 0.0) for n_ in names)        # every tensor read once + written once (engine's ledger)
                avg_s = ms * 1e-3 / n if n else None
                return {"kernel": kernel, "bound": "mfma", "achieved": tf(ex), "peak": peak, "unit": "TFLOP/s",
                        "frac": tf(ex) / peak, "algorithmic": tf(al), "algorithmic_frac": tf(al) / peak,
                        "launches_per_step": n / prof_steps, "ms_per_step": ms / prof_steps,
                        "avg_launch_ms": ms / n if n else None, "issued_flop_per_launch": ex / n if n else None,
                        "alg_flop_per_launch": al / n if n else None, "traffic": tr,
                        # north_star: "rocprof HBM GB/s ... against gfx950 peak".  Counter bytes per launch / this run's average launch
                        # duration; the same for the algorithmic bytes (input + output + filter of every launch, l3_profile_read_bytes)
                        "alg_bytes_per_launch": ab / n if n else None,
                        "alg_hbm_tbs": (ab / n) / avg_s / 1e12 if n and avg_s else None,
                        "hbm_tbs": None if tr is None or not avg_s else tr / avg_s / 1e12,
                        "hbm_frac": None if tr is None or not avg_s else tr / avg_s / 1e12 / PEAK_HBM_TBS,
                        "hbm_frac_of_measured_peak": None if tr is None or not avg_s else tr / avg_s / 1e12 / MEASURED_HBM_TBS,
                        "hbm_peak_tbs": PEAK_HBM_TBS, "hbm_measured_peak_tbs": MEASURED_HBM_TBS,
                        "traffic_over_algorithmic": None if tr is None or not ab else tr / (ab / n),
                        "traffic_stale": None if tr is None else stale(traffic, SRC_OF.get(traffic_key, ())),
                        "traffic_source": None if tr is None else "profiles/pmc_traffic.json (builder-side rocprofv3 --pmc pass of "
                                                                  "this kernel, FETCH_SIZE x2 + WRITE_SIZE; not measured in this run)"}
            if args.dtype == 'f32':
                fams = {"conv_wgrad": family(['conv_wgrad'], "conv_wgrad_wino_kernel<*> (Winograd F(3x3,2x2) weight gradient on "
                                             "v_mfma_f32_32x32x2_f32; incl. the two direct first-layer launches, the split-K "
                                             "reduces and the output transform)", 'conv_wgrad_wino'),
                        "conv_fwd_dgrad": family(['conv_fwd', 'conv_dgrad'], "conv_wino4_kernel<*> (Winograd F(4x4,3x3) forward + data gradient of the 14 "
                                                 "3x3 layers on v_mfma_f32_32x32x2_f32; incl. the two direct first-layer launches)",
                                                 'conv_wino4')}
                if args.fp32_conv == 'f2x2_bf16x6':      # issued flops of this family are bf16 MFMA flops (six products per fp32 one)
                    fams["conv_fwd_dgrad"] = family(['conv_fwd', 'conv_dgrad'], "conv_wino_bx6_kernel<*> (Winograd F(2x2,3x3), fp32 operands as exact "
                                                    "bfloat16 triples, six cross products on v_mfma_f32_32x32x16_bf16; incl. the two direct fp32 "
                                                    "first-layer launches); frac is of the BF16 matrix peak", 'conv_wino_bx6', PEAK_BF16_MFMA_TFLOPS)
                elif args.fp32_conv == 'f2x2':
                    fams["conv_fwd_dgrad"]["kernel"] = "conv_wino_kernel<*> (Winograd F(2x2,3x3) forward + data gradient on v_mfma_f32_32x32x2_f32)"
            else:
                fams = {"conv_wgrad": family(['conv_wgrad'], "conv_wgrad_bf16_tr_kernel (bf16 weight gradient on ds_read_b64_tr_b16 "
                                             "operands, v_mfma_f32_32x32x16_bf16; incl. the two fp32 first-layer launches)",
                                             'conv_wgrad9t_bf16'),
                        "conv_fwd_dgrad": family(['conv_fwd', 'conv_dgrad'], "conv_bf16_halo_kernel<*> (LDS-halo direct forward + data "
                                                 "gradient, v_mfma_f32_32x32x16_bf16; incl. the first-layer FMA kernels)", 'conv_bf16')}
            # the "dominant kernel" = the family with the largest share of the step
            top = max(fams, key=lambda k: fams[k]['ms_per_step'])
            out["roofline"] = dict(fams[top], measured="%d further steps (outside the timed region) with hipEvents around every "
                                   "launch, towers serialised on one stream" % prof_steps,
                                   note="frac = issued MFMA flops / (duration x peak); algorithmic_frac counts direct-convolution "
                                        "flops (SURVEY 8(d)) and exceeds frac only for Winograd (F(2x2,3x3) issues 16/36 of the "
                                        "direct multiplies, F(4x4,3x3) 9/36)")
            if args.dtype == 'f32' and alu:
                # what `frac` cannot show: on gfx950 an fp32 VALU instruction takes ~4.3 cycles of the SIMD's MATRIX time, and these
                # kernels' own transforms are VALU work -- counted in SIMD cycles the fp32 ALUs are this full (builder-side PMC pass)
                out["roofline"]["fp32_alu_occupancy"] = {
                    k: {"matrix_pipe_busy": round(v["matrix_pipe_busy"], 3), "with_own_valu": round(v["alu_busy"], 3),
                        "valu_per_mfma": round(v["valu_per_mfma"], 2)}
                    for k, v in alu.items() if isinstance(v, dict) and 'matrix_pipe_busy' in v and ('conv_wino' in k or 'conv_wgrad_wino' in k)}
                out["roofline"]["fp32_alu_occupancy_stale"] = stale(alu, ['conv_wino4.hip', 'conv_wgrad_wino.hip'])
                out["roofline"]["fp32_alu_occupancy_source"] = "profiles/pmc_alu.json (builder-side rocprofv3 --pmc pass; not measured in this run)"
            out["kernels"] = fams
            out["kernel_ms_per_step"] = {k: v['ms'] / prof_steps for k, v in prof.items()}
            ew = prof['elementwise']['ms'] / prof_steps
            ew_bytes = prof['elementwise'].get('alg_bytes', 0.0) / prof_steps
            ew_tr = traffic.get('elementwise', {}) if isinstance(traffic.get('elementwise'), dict) else {}
            ew_cnt = ew_tr.get('hbm_bytes_per_step')
            out["elementwise"] = {"bound": "hbm", "ms_per_step": ew, "peak": PEAK_HBM_TBS, "measured_peak": MEASURED_HBM_TBS, "unit": "TB/s",
                                  # algorithmic bytes per step = sum over the family's launches of the tensors each pass reads + writes
                                  # (engine ledger, l3_profile_read_bytes); `achieved` = that / the family's serialised time
                                  "alg_bytes_per_step": ew_bytes, "launches_per_step": prof['elementwise']['launches'] / prof_steps,
                                  "achieved": ew_bytes / (ew * 1e-3) / 1e12 if ew > 0 else None,
                                  "frac": ew_bytes / (ew * 1e-3) / 1e12 / PEAK_HBM_TBS if ew > 0 else None,
                                  "frac_of_measured_peak": ew_bytes / (ew * 1e-3) / 1e12 / MEASURED_HBM_TBS if ew > 0 else None,
                                  "traffic": ew_cnt,
                                  "hbm_tbs": None if ew_cnt is None or ew <= 0 else ew_cnt / (ew * 1e-3) / 1e12,
                                  "traffic_over_algorithmic": None if ew_cnt is None or not ew_bytes else ew_cnt / ew_bytes,
                                  "traffic_source": None if ew_cnt is None else "profiles/pmc_traffic.json (builder-side rocprofv3 --pmc pass over "
                                                                                "the BatchNorm / pool kernels of one step; not measured in this run)",
                                  "note": "BatchNorm / ReLU / pool / moving-average kernels (HBM-bound family)"}
        if world == 1 and args.dtype == 'f32' and not args.no_secondary:
            try:
                out["secondary"] = secondary_lines(args, local_rank, tstream)
            except Exception as exc:          # never at the headline's expense
                out["secondary"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, quick=args.quick_cpu_baseline)
            out["cpu_baseline"]["host"] = host_cpu_info()
            out["x_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        json_out.write(json.dumps(out) + '\n')
        json_out.flush()
    ranks.close()
    eng.close()


if __name__ == '__main__':
    main()
