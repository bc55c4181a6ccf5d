import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`-m gpu` on a box without a GPU must FAIL, not pass vacuously by skipping everything (the driver reads the
    exit code).  A plain `pytest tests/` (no -m) on a CPU-only box skips the gpu-marked tests."""
    expr = (config.getoption("-m") or "").strip()
    if "not gpu" in expr:
        return
    if _gpu_present():
        return
    if "gpu" in expr:
        raise pytest.UsageError("-m gpu was requested but no GPU is visible (torch.cuda.is_available() is False): "
                                "the GPU parity tests cannot run here; use gpurun")
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle_py

    oracle_py.build()
    return oracle_py
