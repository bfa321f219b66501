import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


sys.path.insert(0, str(ROOT / "tests"))
from util import usable_cores as _usable_cores  # noqa: E402

# the oracle's OpenMP loops (oracle/libgnn_oracle.so) and the host C++ mirror: before either library starts its runtime
os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cores()))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One gaib context on cuda:0 for the whole GPU test session (fails loudly without the .so)."""
    import torch
    from graphaibench_amd import capi

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    c = capi.Context(0)
    yield c
    c.close()
