#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run: `python tools/prof_summary.py <dir> [top]` -> text table."""
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = sorted(glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {f}\n# total kernel time {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches")
print(f"{'kernel':80s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for r in rows[:top]:
    print(f"{r['Name'][:80]:80s} {r['Calls']:>7s} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:9.2f} "
          f"{float(r['MinNs']) / 1e3:9.2f} {float(r['MaxNs']) / 1e3:9.2f} {float(r['Percentage']):6.2f}")
