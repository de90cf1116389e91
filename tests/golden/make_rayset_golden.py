"""Writes tests/golden/rayset_<config>.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_rayset_golden.py`): the SURVEY 8(d) parity ray sets (tests/rayset.py) and the oracle's
closest hits for them, as hashes of the complete arrays plus a strided subset in full. Like the other golden files
these pin the oracle against itself over time; they are not outputs of the reference (which cannot be built here)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import sro        # noqa: E402
import rayset                 # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

for name, make in rayset.CONFIGS.items():
    desc = make()
    S = sro.OracleScene(desc, build_tables=True)
    o, d, co, cd, par = rayset.oracle_rayset(S, desc)
    hp, cp = S.cast_closest(o, d)
    hc, cc = S.cast_closest(co, cd)
    g = rayset.summary(o, d, co, cd, par, hp, cp, hc, cc)
    np.savez_compressed(os.path.join(OUT, f"rayset_{name}.npz"), **{k: np.asarray(v) for k, v in g.items()})
    print(name, g["n_primary"], g["n_child"], g["primary_hits_sha256"][:12], g["child_hits_sha256"][:12])
