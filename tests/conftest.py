import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as orc
    orc.build()
    return orc


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests/` on a host without a GPU skips the gpu-marked tests instead of failing inside them
    (the driver selects with -m gpu / -m "not gpu"; this is for everybody else)."""
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="needs a GPU (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_device_memory_between_gpu_tests(request):
    """A 1280x720 emulator holds several GB of device scratch: whatever a test left unreachable is collected before the next test
    starts (an emulator caught in a reference cycle would otherwise keep its memory until the cyclic GC happens to run)."""
    yield
    if "gpu" in request.keywords:
        import gc
        gc.collect()
