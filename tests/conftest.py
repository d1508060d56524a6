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


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library must exist for every test tier (built in-tree by __graft_entry__.build())."""
    import __graft_entry__ as g
    lib = os.path.join(ROOT, "recognizing-speech-dysfluencies-in-stuttering_amd", "libdysf_hip.so")
    if not os.path.exists(lib):
        g.build()
    yield
