import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU parity oracle (test infrastructure)."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def lb():
    """The product package; building/loading the HIP library is part of the fixture."""
    import lbaudiodetective_amd as lb
    if not os.path.exists(lb.LIB_PATH):
        lb.build()
    lb.lib()
    return lb


@pytest.fixture(scope="session")
def gpu(lb):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    torch.cuda.set_device(0)
    return torch
