import os
import sys

import pytest

for _k in ("FWD", "BWD", "WRW"):      # keep MIOpen's naive reference solvers out of its first-use solver search
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _k, "0")
os.environ.setdefault("STA_CONV_FIND", "0")   # no per-shape solver measurements here: correctness runs, not speed


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


@pytest.fixture(autouse=True)
def _reset_launch_overrides():
    """Tests may force kernel variants through sta_set_option; never let one leak into the next test."""
    yield
    from sta import lib
    if lib._lib is not None:
        for key in range(9):
            lib._lib.sta_set_option(key, 0)
