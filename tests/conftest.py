import os
os.environ.setdefault('GPU_MAX_HW_QUEUES', '6')   # one hardware queue per HIP stream: detmatch_amd/__init__.py says why (before the runtime initialises)
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _host_conv  # noqa: E402
from detmatch_amd import dense_conv  # noqa: E402

# HOST tensors only: the host logic around the convolutions is checked with torch's own convolution as the
# stand-in (tests/_host_conv.py); CUDA tensors always take the HIP kernels, the product never sets the hook.
dense_conv.HOST_TENSOR_HOOK = _host_conv


# Reference computations of the GPU tests use torch's native kernels, never MIOpen: MIOpen's find step may build a
# kernel at run time (fork + exec of a compiler from a process that already holds the GPU), which aborts the
# test process on hosts that forbid it — and the choice of solver depends on how much memory earlier tests hold.
try:
    import torch
    torch.backends.cudnn.enabled = False
except Exception:      # noqa
    pass


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def orc():
    """The CPU oracle (test infrastructure; never imported by the product)."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
