#!/usr/bin/env python3
"""Static check of the built library for a gfx950 matrix-pipe hazard the compiler does not pad.

Found in round 3: an MFMA whose SrcC is EXACTLY the vDst of an earlier MFMA of a different shape
(v_mfma_f32_16x16x32_bf16 -> v_mfma_f32_16x16x16_bf16: the half k-step of a contraction whose K is not a multiple of 32)
is issued by LLVM (ROCm 7.2) without wait states -- the recogniser treats "same accumulator" as the interlocked
back-to-back case -- but on the MI355X the consumer then reads the accumulator before the producer has written it when
the two are fewer than about ten wait states apart: run-to-run different results (one long-horizon instance, and the peeled
four-sample split-bf16 instance of round 3, traced to it by patching s_nop into the ISA one instruction at a time).

This tool disassembles every gfx950 code object of libbeso_hip*.so and reports each MFMA that takes as SrcC the
result of an MFMA of another opcode with fewer than MIN_WAIT wait states in between (one per instruction, s_nop N = N + 1,
exactly as the compiler's hazard recogniser counts), along every path of the control-flow graph.  Exit code 1 if any.
tests/test_cabi.py runs it on the product library (no GPU needed)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MIN_WAIT = 10          # passes of the slowest producer in this library (8, if v_mfma_f32_16x16x32 is an 8-pass op) + 2
# wait states an instruction between producer and consumer is worth: an independent MFMA keeps the pipe busy for its passes
# (the smallest plausible figures), everything else one issue slot, s_nop N its N + 1
PASSES = {"16x16x32": 4, "32x32x16": 8, "16x16x16": 2, "32x32x8": 4, "16x16x4": 2, "4x4x4": 2}
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    out = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-S", "-W", lib], text=True)
    m = re.search(r"\.hip_fatbin\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", out)
    if not m:
        raise SystemExit(f"{lib}: no .hip_fatbin section")
    off, size = int(m.group(2), 16), int(m.group(3), 16)
    with open(lib, "rb") as f:
        f.seek(off)
        data = f.read(size)
    pos = 0
    while True:
        p = data.find(MAGIC, pos)
        if p < 0:
            return
        n = struct.unpack_from("<Q", data, p + 24)[0]
        q = p + 32
        for _ in range(n):
            o, s, idl = struct.unpack_from("<QQQ", data, q)
            q += 24
            tid = data[q:q + idl].decode()
            q += idl
            if "gfx950" in tid and s:
                yield data[p + o:p + o + s]
        pos = p + 1


def vregs(tok):
    m = re.match(r"^([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return frozenset((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.match(r"^([va])(\d+)$", tok)
    return frozenset({(m.group(1), int(m.group(2)))}) if m else frozenset()


def check_function(name, lines):
    """lines: instructions and '<Lnn>:' labels of one function.  Forward dataflow over the CFG; the state maps an accumulator
    register to (opcode of the MFMA that wrote it last, wait states since), merged by the smaller distance."""
    label = {l[1:-2]: i for i, l in enumerate(lines) if l.startswith("<")}
    found = {}
    state_at = {}
    work = [(0, {})]
    while work:
        i, st = work.pop()
        st = dict(st)
        while i < len(lines):
            l = lines[i]
            if l.startswith("<"):
                old = state_at.get(i)
                if old is not None:
                    merged = dict(old)
                    changed = False
                    for r, (op, age) in st.items():
                        if r not in merged or merged[r][1] > age:
                            merged[r] = (op, age)
                            changed = True
                    if not changed:
                        break
                    st = merged
                state_at[i] = dict(st)
                i += 1
                continue
            parts = l.split(None, 1)
            opc, ops = parts[0], (parts[1] if len(parts) > 1 else "")
            toks = [t for t in re.split(r",\s*|\s+", ops) if t]
            if opc.startswith(("v_mfma", "v_smfmac")):
                dst, srcc = vregs(toks[0]), (vregs(toks[3]) if len(toks) > 3 else frozenset())
                for r in srcc:
                    if r in st and st[r][0] != opc and st[r][1] < MIN_WAIT:
                        found.setdefault(i, (l, st[r][0], st[r][1]))
                step = next((v for k, v in PASSES.items() if k in opc), 2)
                st = {r: (op, age + step) for r, (op, age) in st.items() if age + step < MIN_WAIT}
                for r in dst:
                    st[r] = (opc, 0)
            else:
                step = 1
                if opc == "s_nop":
                    step = int(toks[0], 0) + 1
                # any other instruction writing the register ends the chain
                wr = vregs(toks[0]) if toks and not opc.startswith(("ds_write", "buffer_store", "global_store", "scratch_store")) else frozenset()
                st = {r: (op, age + step) for r, (op, age) in st.items() if age + step < MIN_WAIT and r not in wr}
            if opc == "s_endpgm":
                break
            if opc == "s_branch":
                tgt = re.search(r"<(L\d+)>", ops)
                if not tgt:
                    break
                i = label[tgt.group(1)]
                continue
            if opc.startswith("s_cbranch"):
                tgt = re.search(r"<(L\d+)>", ops)
                if tgt:
                    work.append((label[tgt.group(1)], st))
            i += 1
    return [(name, i, *v) for i, v in sorted(found.items())]


def check_library(lib):
    bad, n_fn, n_mfma = [], 0, 0
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".o") as f:
            f.write(co)
            f.flush()
            dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--symbolize-operands", f.name], text=True)
        name, lines = None, []
        for raw in dis.split("\n") + ["0000 <end>:"]:
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", raw)
            if m and re.fullmatch(r"L\d+", m.group(1)):
                lines.append(f"<{m.group(1)}>:")
                continue
            if m:
                if name and lines:
                    n_fn += 1
                    n_mfma += sum(1 for l in lines if l.startswith("v_mfma"))
                    bad += check_function(name, lines)
                name, lines = m.group(1), []
                continue
            t = raw.split("//")[0].strip()
            if t and not t.startswith(("Disassembly", "/")) and "file format" not in t:
                lines.append(t)
    return bad, n_fn, n_mfma


if __name__ == "__main__":
    libs = sys.argv[1:] or [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "beso_amd", "lib", "libbeso_hip.so")]
    rc = 0
    for lib in libs:
        bad, n_fn, n_mfma = check_library(lib)
        print(f"{os.path.basename(lib)}: {n_fn} kernels, {n_mfma} MFMAs, {len(bad)} mixed-shape accumulator chains closer than {MIN_WAIT} wait states")
        for name, i, ins, prod, age in bad[:40]:
            print(f"  {name[:90]}  +{i}: {ins}   <- {prod} {age} wait states earlier")
        rc |= 1 if bad else 0
    sys.exit(rc)
