#!/usr/bin/env python3
"""Per-kernel hash of the gfx950 ISA of a built library (or object): `kernel_isa_hash.py lib.so > a.txt`, then diff two builds.
Used to check that a source change leaves the instruction streams of kernels it should not touch identical."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_mfma_chains import LLVM, code_objects


def main(lib):
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", f.name], text=True)
        cur, body = None, []
        out = {}
        for line in txt.splitlines():
            m = re.match(r"^<?([_A-Za-z0-9.$]+)>?:$", line.strip())
            if m:
                if cur:
                    out[cur] = body
                cur, body = m.group(1), []
            elif cur:
                body.append(re.sub(r"\s*//.*$", "", line).strip())
        if cur:
            out[cur] = body
        for k in sorted(out):
            if k.startswith("_Z"):
                h = hashlib.sha1("\n".join(out[k]).encode()).hexdigest()[:12]
                print(f"{h} {len(out[k]):6d} {k}")


if __name__ == "__main__":
    main(sys.argv[1])
