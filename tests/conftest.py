"""pytest configuration: the `gpu` marker selects tests that need a real MI355X.

`pytest -m "not gpu"` runs on the CPU-only build container (oracle vs golden vectors, host logic, C-ABI
symbol export, gloo world_size-2 paths); `pytest -m gpu` is the parity suite proper and calls the HIP
kernels through the C ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an AMD MI355X (gfx950) device")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch
    return torch.device("cuda:0")


@pytest.fixture
def kopt():
    """Set libclipk kernel-selection options for one test (include/clipk.h: clipk_set_option); everything is reset
    to the defaults afterwards.  Options never change results, only which kernel / schedule computes them."""
    from clip_dplm_amd import ops

    def _set(name, value):
        ops.set_option(name, value)
    yield _set
    ops.reset_options()
