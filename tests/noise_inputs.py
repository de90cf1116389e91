"""Synthetic framebuffers for the adaptive-sampling controller tests (shared by the CPU and GPU suites)."""
import numpy as np


def framebuffers(W, H, seed=0, samples=70):
    """A frame whose left part is noisy and whose right part has converged, a black (L == 0) patch and a dark band."""
    rng = np.random.default_rng(seed)
    acc = np.zeros((H, W, 4), np.float32)
    aux = np.zeros((H, W, 4), np.float32)
    base = rng.uniform(0.05, 1.5, (H, W, 3)).astype(np.float32)
    rel = np.full((H, W, 1), 0.0015, np.float32)
    rel[:, : W * 2 // 5] = 0.35
    rel[H // 2:, W * 3 // 4:] = 0.08
    acc[..., :3] = base
    aux[..., :3] = base * (1 + rel * rng.normal(size=(H, W, 1))).astype(np.float32)
    acc[H // 4: H // 4 + 5, W // 2: W // 2 + 9, :3] = 0.0          # main luminance 0: skipped by the reference (:571)
    acc[..., 3] = samples
    aux[..., 3] = samples // 2
    acc[3, 5, 3] = samples - 2                                       # the minimum sample count of the frame
    return acc.reshape(-1, 4), aux.reshape(-1, 4)
