#!/usr/bin/env python3
"""Development aid: static instruction mix of the fused layers kernel per barrier-delimited region.
VALU instructions do not hide under MFMAs on gfx950 beyond ~one per MFMA (tools/microbench/issue_rate.hip),
so the VALU count of each phase is a direct cost.   python tools/valu_count.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tmp = tempfile.mkdtemp(prefix="valu_")
    src = os.path.join(ROOT, "beso_amd", "csrc", "fused.hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-save-temps", "-c", src,
           "-o", os.path.join(tmp, "fused.o"), *sys.argv[1:]]
    subprocess.run(cmd, cwd=tmp, check=True, stderr=subprocess.DEVNULL)
    text = open(os.path.join(tmp, "fused-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    # the throughput instance of the kitchen shape: layers_kernel<3, 12, 1, 2, 8, 6, 0>
    sym = os.environ.get("BESO_VALU_SYMBOL", "_ZN4beso12_GLOBAL__N_113layers_kernelILi3ELi12ELi1ELi2ELi8ELi6ELi0EE")
    start = next(i for i, l in enumerate(text) if l.startswith(sym) and ": ;" in l)
    ends = [i for i, l in enumerate(text) if "s_endpgm" in l and i > start]
    lines = text[start:ends[0] + 1]
    bars = [i for i, l in enumerate(lines) if "s_barrier" in l]
    prev = 0
    tot_v = tot_m = 0
    for k, b in enumerate(bars + [len(lines)]):
        v = m = lds = vm = 0
        kinds = {}
        for l in lines[prev:b]:
            t = l.strip().split()
            if not t:
                continue
            op = t[0]
            if op.startswith("v_mfma"):
                m += 1
            elif op.startswith("v_"):
                v += 1
                kk = re.sub(r"_e32|_e64", "", op)
                kinds[kk] = kinds.get(kk, 0) + 1
            elif op.startswith("ds_"):
                lds += 1
            elif op.startswith("global_") or op.startswith("scratch_"):
                vm += 1
        top = ", ".join(f"{a} {c}" for a, c in sorted(kinds.items(), key=lambda x: -x[1])[:7])
        print(f"region {k:2d} [{prev:6d},{b:6d}): VALU {v:5d} MFMA {m:4d} LDS {lds:4d} VMEM {vm:4d} | {top}")
        prev = b
        tot_v += v
        tot_m += m
    print(f"static totals: VALU {tot_v}, MFMA {tot_m}")


if __name__ == "__main__":
    main()
