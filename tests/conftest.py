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
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu():
    """The product on the GPU: fails (does not skip, does not fall back) if the HIP library is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import cudf_amd
    cudf_amd._lib.load()
    return cudf_amd
