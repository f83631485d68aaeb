#!/usr/bin/env python3
"""Readable per-kernel table from a rocprofv3 --stats kernel_stats.csv:  python tools/kernel_stats.py <csv> [steps]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    n = re.sub(r"beso::\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"\(.*", "", n)[:88]
    ms = float(r["TotalDurationNs"]) / steps / 1e6
    tot += ms
    print(f"{n:90s} calls/step {int(r['Calls']) / steps:6.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  per step {ms:6.3f} ms  {r['Percentage']}%")
print(f"sum of the rows shown: {tot:.3f} ms per step")
