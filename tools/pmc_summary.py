#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv files under a directory) into per-kernel means.

    python tools/pmc_summary.py gpurun_out/pmc1 [gpurun_out/pmc2 ...] > profiles/rNN_pmc_x.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = row["Kernel_Name"][:90]
                    c = acc[k][row["Counter_Name"]]
                    c[0] += float(row["Counter_Value"])
                    c[1] += 1
    out = {k: {n: v[0] / v[1] for n, v in sorted(cs.items())} | {"dispatches": max(v[1] for v in cs.values())}
           for k, cs in acc.items()}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
