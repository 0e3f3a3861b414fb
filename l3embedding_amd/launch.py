"""One command, N GPUs.  The reference runs its N replicas inside one process (`python 03_train_embedding.py --gpus 4`
-> train.py:263-267 -> multi_gpu_model); here data parallelism is one process per GPU, so a command started without a
launcher re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1."""
import os
import socket
import sys

SPAWNED_FLAG = 'L3_SPAWNED_UNDER_LAUNCHER'


def launched():
    """True when a launcher (torchrun or this module) already set the rank environment."""
    return 'WORLD_SIZE' in os.environ and 'RANK' in os.environ


def check_world(n, what='--gpus'):
    """A launcher that started another number of ranks than asked for is an error, not a silent smaller run."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != n:
        raise SystemExit('%s %d but the launcher started %d ranks' % (what, n, world))
    return world


def launcher_command(n, target, argv, port=None):
    """`target` = ['path/to/script.py'] or ['-m', 'package.module']."""
    if port is None:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
            '127.0.0.1', '--master-port', str(port)] + list(target) + list(argv)


def respawn_under_launcher(n, target, argv, extra_env=None):
    if os.environ.get(SPAWNED_FLAG):
        raise SystemExit('re-executed under torch.distributed.run but no rank environment arrived')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    env[SPAWNED_FLAG] = '1'
    env.update(extra_env or {})
    os.execvpe(sys.executable, launcher_command(n, target, argv), env)
