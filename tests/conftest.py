import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# bin/paffy hands its sub-commands to another paffy on PATH unless told otherwise: the suite tests the MI355X implementation,
# whatever the environment holds (a test that checks the hand-over sets MIPAF_NATIVE=0 itself)
os.environ["MIPAF_NATIVE"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def olz():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import olz as _olz
    _olz.load()
    return _olz


@pytest.fixture(scope="session")
def gpu_ctx():
    from cactus_amd import miblast
    ctx = miblast.Context(0)          # raises without a gfx950 device: GPU tests must not pass on a fallback
    yield ctx
    ctx.close()
