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
    from oracle import oracle_binding
    oracle_binding.load_oracle()
    return oracle_binding


@pytest.fixture(scope="session")
def lib():
    from cvgpuspeedup_amd import capi
    return capi.load_library()


@pytest.fixture(scope="session")
def device():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")
