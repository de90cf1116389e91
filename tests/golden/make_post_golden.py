"""Writes tests/golden/post_golden.json: the oracle's output stage on 64 seeded pixels (tonemap 0 and 1).
PARITY UNPINNED: the reference's ImageHandler needs OpenColorIO and a config file outside its repository, so these
vectors pin the oracle against itself (a regression fixture), not against the reference."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import sro
from post_inputs import hdr_pixels

px = hdr_pixels(64)
out = {"pixels_u32": px.view(np.uint32).tolist()}
for tm in (0, 1):
    f, u8 = sro.post_process(px, tm)
    out[f"float_u32_tonemap{tm}"] = f.view(np.uint32).tolist()
    out[f"rgba8_tonemap{tm}"] = u8.tolist()
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "post_golden.json"), "w"))
print("wrote", len(px), "pixels")
