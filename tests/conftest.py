import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped on a box without a GPU (a plain `pytest` here then passes) unless FOVRASTER_REQUIRE_GPU=1,
    which turns the absence into a failure (the -m gpu run on the MI355X box must never silently skip)."""
    if os.environ.get("FOVRASTER_REQUIRE_GPU") == "1" or _gpu_present():
        return
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        return  # an explicit -m gpu run without a GPU fails loudly inside the tests
    skip = pytest.mark.skip(reason="needs the MI355X (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_runtest_setup(item):
    # comparisons recorded by a test marked `gpu` go to the GPU report, all others to the CPU one (tests/parity_report.py)
    from tests import parity_report
    parity_report.current_test_is_gpu = "gpu" in item.keywords
    if "gpu" in item.keywords:
        # every gradient tensor of a GPU test starts as NaN: fr_backward writes all of it (rows AND zeros) or the comparison fails
        import importlib
        importlib.import_module("fov3dgs_amd")
        from fov3dgs_amd import rasterizer
        rasterizer.POISON_GRADIENTS = True


def pytest_sessionfinish(session, exitstatus):
    try:
        from tests import parity_report
        parity_report.flush()
    except Exception:
        pass
