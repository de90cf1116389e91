import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tables():
    """(arccos, pdf) lookup tables built once by the oracle (PDF::prep restated, ~0.2 s)."""
    from oracle import sro
    from raytracing_amd import scenes
    S = sro.OracleScene(scenes.cornell(16, 1), build_tables=True)
    return S.tables()


def random_rays(n, seed, extent=4.0):
    """Incoherent rays: origins in a box, unit directions (plus a few axis-aligned ones -> inf/NaN slabs)."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-extent, extent, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    k = min(n // 8, 64)
    for i in range(k):
        d[i] = 0
        d[i, i % 3] = 1.0 if (i // 3) % 2 == 0 else -1.0
    for i in range(k, 2 * k):
        d[i, i % 3] = 0.0
    return o, d
