import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name, dtype):
    return np.fromfile(os.path.join(GOLDEN, name), dtype=dtype)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/pm_oracle.c) -- the checker, never the thing under test on the GPU."""
    from oracle import orc_py
    orc_py.lib()
    return orc_py


@pytest.fixture(scope="session")
def gpu_ctx():
    import raytracer_amd as ra
    ctx = ra.Context(0)
    ctx.upload_scene(ra.default_scene())
    yield ctx
    ctx.close()


def psnr(a, b):
    """PSNR over the three colour channels of two 0xAARRGGBB panes."""
    a = np.asarray(a, dtype=np.uint32).ravel(); b = np.asarray(b, dtype=np.uint32).ravel()
    sse = 0.0
    for sh in (0, 8, 16):
        d = ((a >> sh) & 0xFF).astype(np.float64) - ((b >> sh) & 0xFF).astype(np.float64)
        sse += float((d * d).sum())
    mse = sse / (3 * a.size)
    return float("inf") if mse == 0 else 10 * np.log10(255.0 * 255.0 / mse)
