#!/usr/bin/env python3
"""Per-kernel, per-launch averages of rocprofv3 --pmc passes over tools/bench_train.py (run by tools/record_train.sh):
    python tools/pmc_train.py gpurun_out/pmc_train_a gpurun_out/pmc_train_b ... > profiles/r01_train_step_pmc.json"""
import collections
import csv
import glob
import json
import re
import sys

out = collections.defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        sums, cnts = collections.defaultdict(float), collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            n = re.sub(r"beso::\(anonymous namespace\)::", "", r["Kernel_Name"])
            n = re.sub(r"\(.*", "", n).replace("void ", "")
            if not n.startswith(("tgemm", "ln_", "attn_", "colsum", "beso::adam", "train_embed", "train_fwd", "train_pack", "pack_table", "train_dgrad", "train_mlp_bwd",
                                 "slab_reduce", "wgrad_reduce", "wgrad_panel")):
                continue
            sums[(n, r["Counter_Name"])] += float(r["Counter_Value"])
            cnts[(n, r["Counter_Name"])] += 1
        for (n, c), v in sums.items():
            out[n][c] = v / cnts[(n, c)]
            out[n]["_launches_" + c] = cnts[(n, c)]
json.dump(out, sys.stdout, indent=1, sort_keys=True)
