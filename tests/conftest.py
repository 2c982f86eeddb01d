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


class heartbeat:
    """Keeps a long CPU-side step of a GPU test (the oracle's brute-force scan of 10^7 triangles runs for a minute or two)
    from looking hung to a harness that watches for output: while the block runs, a thread rewrites a small file under
    gpurun_out/ every 20 s.  (pytest captures stdout/stderr, so printing would not be seen.)"""

    def __init__(self, what):
        self.what = what

    def __enter__(self):
        import threading
        import time
        self._stop = threading.Event()
        self._t0 = time.time()
        path = os.path.join(ROOT, "gpurun_out")

        def beat():
            n = 0
            while not self._stop.wait(20.0):
                n += 1
                try:
                    os.makedirs(path, exist_ok=True)
                    with open(os.path.join(path, "heartbeat.txt"), "w") as f:
                        f.write("%s: %d s\n" % (self.what, int(time.time() - self._t0)))
                except OSError:
                    pass

        self._thread = threading.Thread(target=beat, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join()
        return False
