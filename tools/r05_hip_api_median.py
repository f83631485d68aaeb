#!/usr/bin/env python3
"""Median host duration and calls per step of every HIP API in a rocprofv3 --hip-trace CSV of tools/bench_train.py (steady state:
the last 60 % of the calls of each API):   python tools/r05_hip_api_median.py <dir> <steps>"""
import collections, csv, glob, statistics, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    dur[r["Function"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for k, v in dur.items():
    tail = v[int(len(v) * 0.4):]
    rows.append((statistics.median(tail) * len(v) / steps / 1000.0, k, len(v) / steps, statistics.median(tail) / 1000.0))
for tot, k, n, med in sorted(rows, reverse=True)[:14]:
    print(f"{k:34s} calls/step {n:7.1f}  median {med:7.2f} us  ~{tot:8.1f} us/step")
