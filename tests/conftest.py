import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _finalize_between_tests():
    """Objects of a finished test that own HIP resources (captured graphs and their private pools,
    pinned staging rings, events) are collected HERE, at a safe point -- not whenever the
    collector happens to run inside the next test, possibly in the middle of a stream capture,
    where a finalizer's HIP call is an error thrown from a destructor (process abort)."""
    yield
    import gc

    gc.collect()
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available():
        torch.cuda.synchronize()
