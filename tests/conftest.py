import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

import detmatch_amd  # noqa: E402,F401  (sets the MIOpen environment before the first convolution)
from detmatch_amd import dense_conv  # noqa: E402

# CPU tensors only: host logic around the convolutions is checked with torch's own convolution as the
# stand-in; CUDA tensors always take the HIP kernels (the product never sets this flag).
dense_conv.TORCH_REFERENCE_FOR_TESTS = True


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
