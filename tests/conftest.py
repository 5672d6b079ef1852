import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


@pytest.fixture(scope="session")
def oracle():
    from oracle import lfd_oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    """One native context for the whole GPU session.  Fails (never skips to a CPU path) when the
    HIP library or the device is missing."""
    from lfd_amd import _native
    ctx = _native.Context(0, 1489, 2048, 8)
    yield ctx
    ctx.close()


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
