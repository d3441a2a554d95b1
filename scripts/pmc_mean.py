#!/usr/bin/env python3
"""dev: mean value per dispatch of every counter in a rocprofv3 --pmc output directory, for kernels matching a substring.
usage: pmc_mean.py <dir> <kernel substring>"""
import csv, sys
from collections import defaultdict
from pathlib import Path

root, sub = Path(sys.argv[1]), sys.argv[2]
acc = defaultdict(lambda: [0.0, 0])
for f in root.rglob("*counter_collection.csv"):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if sub in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
for name, (tot, cnt) in sorted(acc.items()):
    print(f"{name:44s} dispatches {cnt:6d}  mean {tot / cnt:16.1f}")
