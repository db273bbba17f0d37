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


def pytest_collection_modifyitems(config, items):
    """A host without a GPU SKIPS the device tests instead of failing them (someone running the
    whole directory on a CPU box; the drivers select with -m gpu / -m "not gpu")."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionstart(session):
    """The library under test is the one built in-tree: a process-wide override of its path
    (PFRL_AMD_LIB, honoured by pfrl_amd/_native.py for debug builds) would make every parity
    claim of this suite about some other binary."""
    if os.environ.get("PFRL_AMD_LIB"):
        raise pytest.UsageError("PFRL_AMD_LIB is set (%s): the test-suite validates "
                                "pfrl_amd/lib/libpfrl_amd.so only" % os.environ["PFRL_AMD_LIB"])


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
