#!/usr/bin/env python3
"""VGPRs / spills / scratch per kernel from `hipcc ... -Rpass-analysis=kernel-resource-usage 2> remarks.txt`:
    python tools/resource_usage.py remarks.txt [name-filter-regex]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    dem = subprocess.check_output(["c++filt", name], text=True).strip()
    dem = re.sub(r"beso::\(anonymous namespace\)::", "", dem)
    dem = re.sub(r"\(.*", "", dem)
    if flt and not flt.search(dem):
        continue
    g = lambda k: re.search(re.escape(k) + r": (\d+)", b).group(1)      # noqa: E731
    print("%-56s VGPR %3s AGPR %3s spill %3s scratch %4s occ %s" % (dem[:56], g("VGPRs"), g("AGPRs"), g("VGPRs Spill"),
                                                                   g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]")))
