import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree HIP library; built on demand (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    from g1_locomotion_amd import _lib
    return _lib.load()
