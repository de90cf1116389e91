#!/usr/bin/env python3
"""Writes tests/golden/noise_golden.json: the oracle's region list and a sample of block priorities for the
synthetic framebuffers of tests/noise_inputs.py (pins the controller restatement against itself over time)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import sro                       # noqa: E402
from noise_inputs import framebuffers        # noqa: E402

W, H, FB_SEED, SEED, ANALYSES, STRIDE = 160, 96, 3, 11, 2, 37
acc, aux = framebuffers(W, H, seed=FB_SEED)
err, mn = sro.noise_error_map(acc, aux)
N = sro.OracleNoise(W, H, seed=SEED)
for _ in range(ANALYSES):
    N.analyze(err, mn)
rects, stats = N.regions()
pr, _ = N.priorities(mn)
out = {"W": W, "H": H, "fb_seed": FB_SEED, "seed": SEED, "analyses": ANALYSES, "rects": rects.tolist(),
       "active": stats[:, 0].astype(int).tolist(), "noise": [float(v) for v in stats[:, 1]],
       "priority_stride": STRIDE, "priority_sample": [float(v) for v in pr[::STRIDE]]}
with open(os.path.join(HERE, "noise_golden.json"), "w") as fh:
    json.dump(out, fh)
print(len(rects), "regions")
