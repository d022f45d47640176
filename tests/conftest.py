import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """`pytest` without `-m` on a box with NO GPU: gpu-marked tests are skipped, not failed.  On a GPU box nothing is skipped:
    a missing or stale libcfpnet_hip.so must fail loudly there."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="gpu test: no GPU on this box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
