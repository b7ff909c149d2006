#!/usr/bin/env python3
"""Per-kernel mean of a PMC counter from a rocprofv3 --pmc run: `python tools/pmc_summary.py <dir>`."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        acc[name.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    for c, v in cs.items():
        print(f"{k:62s} {c:12s} n={len(v):5d} mean={sum(v) / len(v):14.1f}")
