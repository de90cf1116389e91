#!/usr/bin/env python3
"""CPU experiment (oracle only): does the product's leaf filter ever skip a leaf in which the reference's fp32
triangle test accepts a hit? Counts, over adversarial scenes and rays, the closest-hit updates the unfiltered
reference walk takes inside leaves the filter would skip (must be 0 for the filter to be exact)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sro                      # noqa: E402
from raytracing_amd import scenes           # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from adversarial import ADVERSARIAL_SCENES, adversarial_rays   # noqa: E402


def main():
    total = 0
    for name, make in ADVERSARIAL_SCENES.items():
        desc = make()
        S = sro.OracleScene(desc, build_tables=False)
        for kind, (o, d) in adversarial_rays(desc, seed=5).items():
            with np.errstate(all="ignore"):
                c, ctr = S.cast_closest(o, d)
            hits = int((c["material"] != 0xFFFFFFFF).sum())
            print(f"{name:18s} {kind:18s} rays {len(o):7d} hits {hits:7d} packets {int(ctr[1]):9d} filtered {int(ctr[3]):9d} "
                  f"guarded {int(ctr[6]):8d} accepts-in-filtered {int(ctr[4]):5d} passes-in-filtered {int(ctr[5]):5d}", flush=True)
            total += int(ctr[4])
    print("TOTAL accepts in filtered leaves:", total)


if __name__ == "__main__":
    main()
