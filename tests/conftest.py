import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` via gpurun)")


@pytest.fixture(scope="session")
def hip_lib():
    """liblara2dgs.so built in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
    import __graft_entry__ as g
    from lara_amd import rasterizer
    if not os.path.exists(rasterizer.LIB_PATH):
        g.build()
    return rasterizer.load_library()


@pytest.fixture(autouse=True)
def _poison_guards_survive(request):
    """With LARA2DGS_POISON_BUFFERS=1 in the environment of the whole run (a debugging mode: every state / scratch buffer of
    the rasteriser 0xFF-filled between guard zones), every GPU test also asserts that no kernel wrote outside a buffer."""
    yield
    if os.environ.get("LARA2DGS_POISON_BUFFERS") == "1" and request.node.get_closest_marker("gpu") is not None:
        from lara_amd import rasterizer
        assert rasterizer.check_poison_guards() == [], "a kernel wrote beyond the end of a state / scratch buffer"
