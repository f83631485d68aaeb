#!/usr/bin/env python3
"""The vector-memory skeleton of one kernel of a built object: loads, stores, waits, branches -- to see how many loads the
compiler really keeps in flight:  python tools/kernel_loads.py beso_amd/build/train.o ln_reduce_kernel [max_lines]"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_mfma_chains import LLVM, code_objects

obj, name = sys.argv[1], sys.argv[2]
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 60
for co in code_objects(obj):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co)
        f.flush()
        txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", f.name], text=True)
    for m in re.finditer(r"^<?(\S*%s\S*)>?:$" % re.escape(name), txt, re.M):
        a = m.end()
        b = txt.find("s_endpgm", a)
        print("==", m.group(1))
        n = 0
        for line in txt[a:b].split("\n"):
            t = re.sub(r"\s*//.*$", "", line).strip()
            if re.match(r"(global_load|global_store|buffer_load|buffer_store|scratch_|s_waitcnt vmcnt|s_cbranch|s_barrier)", t):
                print("  ", t)
                n += 1
                if n >= cap:
                    break
