import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores():
    """CPU threads this process may really use (affinity mask, cgroup quota), capped at the GPU box's 16-per-GPU share: some boxes show
    all 256 host CPUs to a 16-core allotment, and torch's default of 128 intra-op threads then makes the CPU oracle 5-10x slower."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cores()))
    try:
        import torch
        torch.set_num_threads(_usable_cores())
    except Exception:
        pass


@pytest.fixture(scope="session")
def gpu_ctx():
    """Context on torch's current stream of cuda:0. Fails (not skips) when the HIP library cannot be used."""
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    from ampis_amd import ops
    ctx = ops.torch_context(0)
    yield ctx
    ctx.close()
