import os
os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')   # the runtime's default, pinned: with RCCL initialised 5+ hardware queues cost +30 ms per iteration (detmatch_amd/__init__.py)
import sys

# The GPU tests run the iteration in the order that SHIPS (three stream lanes, chains, early issue: the bench default)
# unless a test sets another one itself.  (Round 5 ran them in the one-lane order because of an intermittent device
# dead-lock of the lanes; its cause — two vendor Stream-K GEMMs in flight at once — is removed)

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


def pytest_collection_modifyitems(config, items):
    """A GPU test that wedges must end the run with a failure, not hang it (pytest-timeout, when installed)."""
    try:
        import pytest_timeout  # noqa: F401
    except Exception:      # noqa
        return
    for item in items:
        if item.get_closest_marker('gpu') is not None and item.get_closest_marker('timeout') is None:
            item.add_marker(pytest.mark.timeout(900))


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
