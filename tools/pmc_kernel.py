#!/usr/bin/env python3
"""Per-launch counter averages of ONE kernel from rocprofv3 --pmc output directories:
    python tools/pmc_kernel.py "<kernel name substring>" dir1 [dir2 ...] > profiles/rNN_<what>_pmc.json"""
import collections
import csv
import glob
import json
import sys

sub = sys.argv[1]
out = {"kernel_substring": sub}
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        sums, cnts = collections.defaultdict(float), collections.defaultdict(int)
        name = None
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                name = r["Kernel_Name"]
                sums[r["Counter_Name"]] += float(r["Counter_Value"])
                cnts[r["Counter_Name"]] += 1
        for c, v in sums.items():
            out[c] = v / cnts[c]
            out["_launches_" + c] = cnts[c]
        if name:
            out["kernel"] = name.split("(")[0]
json.dump(out, sys.stdout, indent=1, sort_keys=True)
