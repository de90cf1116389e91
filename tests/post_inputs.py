"""Seeded inputs shared by the output-stage tests (CPU oracle checks and GPU parity)."""
import numpy as np


def hdr_pixels(n, seed=5):
    """Linear RGB pixels spanning 12 stops, with the special cases the stage branches on put first."""
    rng = np.random.default_rng(seed)
    special = np.array([
        [0, 0, 0], [0.5, 0.5, 0.5], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1], [1, 0, 1],
        [0.18, 0.18, 0.18000001], [1e-6, 2e-6, 3e-6], [4, 3, 2], [100, 0.01, 0.01], [0.2, 0.2000001, 0.2],
        [-0.1, 0.5, 0.5], [-1, -1, -1], [np.nan, 0.5, 0.5], [np.inf, 1, 1], [0.3, 0.6, 0.3], [0.6, 0.3, 0.3],
        [0.3, 0.3, 0.6], [1e-3, 0, 0], [0, 0, 1e-12],
    ], np.float32)
    rnd = (2.0 ** rng.uniform(-9, 3, (n - len(special), 1)) * rng.uniform(0, 1, (n - len(special), 3))).astype(np.float32)
    return np.concatenate([special, rnd])[:n]
