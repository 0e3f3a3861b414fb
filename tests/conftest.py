import os
import sys

import pytest

# the kernel-variant switches (L3_WG_WINO, L3_BF16_HALO, ...) some tests flip are honoured only under this gate
# (l3embedding_amd/csrc/knobs.h); the product runs without it
os.environ.setdefault('L3_DEBUG_KNOBS', '1')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope='session')
def gpu_required():
    if not _have_gpu():
        pytest.skip('no GPU visible')
    return True


def need_experiments():
    """Tests of the measured-and-rejected kernel variants (split-bf16 fp32 convolutions, flat-tile MODE 5, tap-split bf16 weight
    gradient) run only against a library built with L3_BUILD_EXPERIMENTS=1 (l3embedding_amd/_build.py): the product library
    does not carry those kernels."""
    from l3embedding_amd import _lib
    if not _lib.experiments_built():
        pytest.skip('libl3hip.so was built without the experiment kernels (L3_BUILD_EXPERIMENTS=1 builds them)')
