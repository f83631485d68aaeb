"""Build libbeso_hip.so (gfx950) in-tree with hipcc.  ``python -m beso_amd.build [--force]``.

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the repo snapshot
(it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(PKG, "build")
LIB = os.path.join(LIBDIR, "libbeso_hip.so")
STAMP = os.path.join(LIBDIR, "libbeso_hip.sha256")        # source hash the .so was built from (travels with it)
# the development build: the same sources with -DBESO_DEV_API=1 (include/beso_hip_debug.h: phase stamps, the GEMM layout
# probe).  Nothing in the package loads it; tools/ and one operand-layout test do.
DEV_LIB = os.path.join(LIBDIR, "libbeso_hip_dev.so")
DEV_STAMP = os.path.join(LIBDIR, "libbeso_hip_dev.sha256")
UNITS = ["api", "elementwise", "attention", "gemm", "fused", "fused_f16", "optim", "train", "feed", "small"]
ARCH = "gfx950"
# -fvisibility=hidden: the dynamic symbol table is exactly include/beso_hip.h (its declarations carry default visibility)
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _source_hash() -> str:
    """sha256 over every file the library is built from (csrc/*, include/beso_hip.h), the flags and the compiler path:
    the key that decides whether the prebuilt .so is current.  (Modification times are not: the .so is git-ignored but
    travels with the tree, and a checkout or copy can make a stale binary look newer than its sources.)"""
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC))
    files.append(os.path.join(os.path.dirname(PKG), "include", "beso_hip.h"))
    files.append(os.path.join(os.path.dirname(PKG), "include", "beso_hip_debug.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS + os.environ.get("BESO_EXTRA_HIPCC_FLAGS", "").split() + UNITS).encode())
    return h.hexdigest()


def _compile(unit: str, dev: bool = False) -> str:
    src = os.path.join(CSRC, unit + ".hip")
    obj = os.path.join(OBJDIR, unit + ("_dev.o" if dev else ".o"))
    extra = os.environ.get("BESO_EXTRA_HIPCC_FLAGS", "").split() + (["-DBESO_DEV_API=1"] if dev else [])
    cmd = [_hipcc(), *FLAGS, *extra, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {unit}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True, dev: bool = False) -> str:
    """Build the product library (dev=False) or the development build (dev=True); returns its path."""
    LIB, STAMP = (DEV_LIB, DEV_STAMP) if dev else (globals()["LIB"], globals()["STAMP"])
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    want = _source_hash()
    try:
        have = open(STAMP).read().strip()
    except OSError:
        have = ""
    if not force and os.path.exists(LIB) and have == want:
        return LIB
    try:
        _hipcc()
    except RuntimeError:
        # no compiler on this box (e.g. a runtime-only image): the shipped binary is all there is; say what it is
        if os.path.exists(LIB):
            sys.stderr.write(f"beso_amd.build: hipcc not found; using the prebuilt library (source hash "
                             f"{have[:12] if have else 'unknown: no stamp file'}, tree is {want[:12]})\n")
            return LIB
        raise
    with cf.ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        objs = list(ex.map(lambda u: _compile(u, dev), UNITS))
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(STAMP, "w") as f:
        f.write(want + "\n")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB, sources {want[:12]})")
    return LIB


def is_current() -> bool:
    """True when the library in lib/ was built from exactly the sources in the tree."""
    try:
        return os.path.exists(LIB) and open(STAMP).read().strip() == _source_hash()
    except OSError:
        return False


if __name__ == "__main__":
    build(force="--force" in sys.argv, dev="--dev" in sys.argv)
