import os
import sys

import pytest

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")     # skip MIOpen's exhaustive solver benchmarking on fresh boxes

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "diffusion-spacetime-attn_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
