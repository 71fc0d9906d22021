import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """The HIP engine on cuda:0 — fails loudly if librt_hip.so is missing or no GPU is present."""
    from raytracing_amd import engine as eng

    eng.init(0)
    yield eng
    eng.shutdown()
