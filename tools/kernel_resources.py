#!/usr/bin/env python3
"""VGPRs / scratch / occupancy of the kernels of one unit:  python tools/kernel_resources.py fused [pattern] [-DFLAG ...]"""
import re
import subprocess
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
flags = [a for a in sys.argv[2:] if a.startswith("-")]
if flags:
    flags.append("-DBESO_VARIANTS=1")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-function",
       *flags, "-c", f"{ROOT}/beso_amd/csrc/{unit}.hip", "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"]
err = open(os.environ["KR_CACHE"]).read() if os.environ.get("KR_CACHE") else subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur.replace("(anonymous namespace)::", "")).replace("beso::", "").replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark: +(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|TotalSGPRs|SGPRs Spill|VGPRs Spill): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if pat in k:
        print(f"{k:70s} VGPR {v.get('VGPRs', 0):4d} AGPR {v.get('AGPRs', 0):4d} SGPR {v.get('TotalSGPRs', 0):4d} scratch {v.get('ScratchSize [bytes/lane]', 0):5d} "
              f"spills v {v.get('VGPRs Spill', 0)} s {v.get('SGPRs Spill', 0)} occ {v.get('Occupancy [waves/SIMD]', 0)}")
