import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The C oracle, compiled on demand (gcc, < 1 s)."""
    from oracle import msmv_oracle
    msmv_oracle.build()
    return msmv_oracle


@pytest.fixture(autouse=True)
def poisoned_gpu_state(request):
    """Before every GPU test: leave NaN in the LDS of every CU (a tiled transpose of a NaN
    tensor) and in the caching allocator's free blocks (so that a later torch.empty hands out
    NaN-filled memory).  A kernel that multiplies stale LDS by zero, or a path that reads an
    output it was supposed to write first, then fails its parity check instead of passing by luck."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    if torch.cuda.is_available():
        from opus_amd import opus_sampling
        block = torch.full((1, 8 * 6, 4 * 64, 64, 176), float("nan"), device="cuda")   # 0.55 GB
        opus_sampling.relayout_features([block], 8, 6, 4)
        del block
        torch.cuda.synchronize()
    yield
