import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import __graft_entry__ as graft  # noqa: E402

graft.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    """device_count() does not initialise the GPU: tests/test_distributed.py spawns worker processes, which must never
    happen from a process that has touched the GPU (the pool refuses such an exec)."""
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device("cuda:0")
