#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel and counter. Usage: pmc_summary.py <dir> [...]"""
import csv, glob, sys
from collections import defaultdict

def main():
    for d in sys.argv[1:]:
        for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
            tot = defaultdict(float); calls = defaultdict(set)
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = r["Kernel_Name"].split("(")[0]
                    tot[(k, r["Counter_Name"])] += float(r["Counter_Value"])
                    calls[k].add(r["Dispatch_Id"])
            print("#", f)
            for (k, c), v in sorted(tot.items()):
                if k.startswith(("k_", "void k_")):
                    print(f"{k:24s} {c:24s} total {v:.6g}  per_dispatch {v / len(calls[k]):.6g}  dispatches {len(calls[k])}")

if __name__ == "__main__":
    main()
