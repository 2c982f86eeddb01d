import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def libs_built():
    """The product libraries are built beforehand by __graft_entry__.build(); tests never compile HIP code."""
    import raytracer_amd.capi as capi
    assert os.path.exists(capi.LIB_HIP) and os.path.exists(capi.LIB_HOST), "run `python -c 'import __graft_entry__ as g; g.build()'`"
    return capi
