import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """Context on torch's current stream of cuda:0. Fails (not skips) when the HIP library cannot be used."""
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    from ampis_amd import ops
    ctx = ops.torch_context(0)
    yield ctx
    ctx.close()
