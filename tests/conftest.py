import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """Product library, initialised on cuda:0.  Fails loudly (no CPU fallback)."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from restir_amd import capi
    capi.init(0)
    torch.zeros(1, device="cuda")
    return capi
