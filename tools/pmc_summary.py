#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/record_profiles.sh into the files kept under profiles/.

    python tools/pmc_summary.py r01 --out gpurun_out/profiles_r01

Writes <tag>_bench.json (the bench line), <tag>_bench_kernel_stats.csv (rocprofv3 --stats per-kernel
table), <tag>_layers_kernel_pmc.json (per-launch counter averages of the dominant kernel) and
traffic.json (HBM bytes per launch = FETCH_SIZE*2 [gfx950 correction, MI355X_MICROARCH.md "HBM"] +
WRITE_SIZE, both reported by rocprofv3 in KiB) which bench.py quotes as roofline.traffic.
"""
import argparse
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_avgs(path, kernel_substr):
    sums, cnts = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel_substr in row["Kernel_Name"]:
                sums[row["Counter_Name"]] += float(row["Counter_Value"])
                cnts[row["Counter_Name"]] += 1
    return {k: sums[k] / cnts[k] for k in sums}, (max(cnts.values()) if cnts else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles"))
    ap.add_argument("--kernel", default="layers_kernel")
    ap.add_argument("--site", default="fused_layer")
    ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    tag, src = a.tag, a.src

    bench = None
    bpath = os.path.join(src, f"bench_{tag}.json")
    if os.path.exists(bpath):
        for line in open(bpath):
            line = line.strip()
            if line.startswith("{"):
                bench = json.loads(line)
        if bench:
            json.dump(bench, open(os.path.join(a.out, f"{tag}_bench.json"), "w"), indent=1)
    stats = os.path.join(src, f"prof_{tag}_stats", f"{tag}_kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(a.out, f"{tag}_bench_kernel_stats.csv"))

    pmc = {}
    for leg in ("fetch", "write", "sq"):
        p = os.path.join(src, f"prof_{tag}_{leg}", f"{tag}_counter_collection.csv")
        if os.path.exists(p):
            avgs, n = counter_avgs(p, a.kernel)
            pmc.update(avgs)
            pmc[f"_launches_{leg}"] = n
    if pmc:
        json.dump(pmc, open(os.path.join(a.out, f"{tag}_layers_kernel_pmc.json"), "w"), indent=1)
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        hbm = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        cfgname = "kitchen"
        tj = {"kernel": a.site, "batch": bench["config"]["batch_per_gpu"] if bench else 4096, "config": cfgname,
              "hbm_bytes_per_launch": hbm,
              "FETCH_SIZE_KiB_avg": pmc["FETCH_SIZE"], "WRITE_SIZE_KiB_avg": pmc["WRITE_SIZE"],
              "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, averaged over the launches of "
                      f"{a.kernel}; read bytes doubled per the gfx950 correction; tag {tag}"}
        json.dump(tj, open(os.path.join(a.out, "traffic.json"), "w"), indent=1)
    print(json.dumps({"bench": bool(bench), "pmc": sorted(pmc)}))


if __name__ == "__main__":
    sys.exit(main())
