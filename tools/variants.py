#!/usr/bin/env python3
"""Development aid: build and time compile-time variants of the fused kernel in one GPU call.

    python tools/variants.py build  name1=-DFLAG=1,-DX=2  name2=@<git-rev>,...   (CPU box: hipcc cross-compile;
                                                                       "@rev" = all of csrc/ as of that commit)
    python tools/variants.py time   [--batch 4096] [--stamps]           (GPU box: times every built variant)

Variants are libbeso_hip_<name>.so under beso_amd/lib/variants/ (only fused.hip is recompiled; the other
objects come from the regular build).  `time` runs each variant in its own process (BESO_HIP_LIB).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "beso_amd", "lib", "variants")


def build(specs):
    from beso_amd import build as B
    B.build()
    os.makedirs(VDIR, exist_ok=True)
    for f in os.listdir(VDIR):
        os.remove(os.path.join(VDIR, f))
    jobs = []
    for spec in specs:
        name, _, flags = spec.partition("=")
        flags = [f for f in flags.split(",") if f]
        rev = [f for f in flags if f.startswith("@")]
        flags = [f for f in flags if not f.startswith("@")]
        if any(f.startswith("-DBESO_") and not f.startswith("-DBESO_DEV_API") for f in flags):
            flags.append("-DBESO_VARIANTS=1")          # (csrc/common.h: the product sources refuse variant macros without it)
        if rev:
            # "@<git revision>": the whole csrc/ + include/ tree as of that commit (A/B against history on one GPU
            # box even when the internal interfaces between the units changed since)
            tree = os.path.join(B.OBJDIR, f"src_{name}")
            subprocess.check_call(["rm", "-rf", tree])
            os.makedirs(tree)
            subprocess.check_call(f"git -C {ROOT} archive {rev[0][1:]} beso_amd/csrc include | tar -x -C {tree}", shell=True)
            csrc = os.path.join(tree, "beso_amd", "csrc")
            units = sorted(f[:-4] for f in os.listdir(csrc) if f.endswith(".hip"))
        else:
            every = "-DBESO_DEV_API=1" in flags or any(f.startswith("-DBESO_TGEMM") for f in flags)     # flags of train.hip too
            train_only = not every and any(f.startswith("-DBESO_WG_") for f in flags)                 # ... of train.hip alone
            every = every or train_only
            csrc, units = B.CSRC, (["train"] if train_only else list(B.UNITS) if every else ["fused", "fused_f16"])
        procs = []
        for u in units:
            obj = os.path.join(B.OBJDIR, f"{u}_{name}.o")
            # (-DBESO_DEV_API=1 changes the entry points of api.hip / train.hip too: such a variant rebuilds every unit)
            extra = flags if (u in ("fused", "fused_f16") or not rev and every) else []
            cmd = [B._hipcc(), *B.FLAGS, *extra, "-c", os.path.join(csrc, u + ".hip"), "-o", obj]
            procs.append((u, obj, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
        jobs.append((name, bool(rev), procs))
    for name, is_rev, procs in jobs:
        objs = []
        for u, obj, p in procs:
            _, err = p.communicate()
            if p.returncode:
                raise SystemExit(f"variant {name}/{u}: hipcc failed\n{err}")
            objs.append(obj)
        if not is_rev:
            built = {u for u, _, _ in procs}
            objs += [os.path.join(B.OBJDIR, u + ".o") for u in B.UNITS if u not in built]
        lib = os.path.join(VDIR, f"libbeso_hip_{name}.so")
        subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib, *objs])
        print("built", lib)


def time_one(batch, steps=300):
    import torch
    from bench import build_model
    from beso_amd import synthetic as O
    dev = "cuda:0"
    cfg = O.SHAPES["kitchen"]
    model = build_model(cfg, O.make_weights(cfg, seed=0, std=0.02), os.environ.get("BESO_VARIANT_PRECISION", "bf16"), dev)
    s, g, a = (torch.from_numpy(v).to(dev) for v in O.make_inputs(cfg, batch, seed=1))
    sig = torch.full((batch,), 0.3, device=dev)
    inner = model.inner_model
    rt, packed = inner.runtime(cfg.sigma_data), inner.packed_weights()
    with torch.no_grad():
        for _ in range(300):                  # a cold chip needs ~100 ms of load to reach its sustained clock
            rt.denoise(packed, s, a, g, sig, precondition=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            rt.denoise(packed, s, a, g, sig, precondition=True)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    if sys.argv[1] == "build":
        return build(sys.argv[2:])
    if sys.argv[1] == "one":
        print(json.dumps({"ms": time_one(int(sys.argv[2]))}))
        return
    batch = 4096
    if "--batch" in sys.argv:
        batch = int(sys.argv[sys.argv.index("--batch") + 1])
    libs = sorted(f for f in os.listdir(VDIR) if f.endswith(".so"))
    for rep in range(2):
        for f in libs:
            env = dict(os.environ, BESO_HIP_LIB=os.path.join(VDIR, f))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(batch)], env=env,
                               capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            ms = json.loads(line[-1])["ms"] if line else None
            print(f"{f[len('libbeso_hip_'):-3]:24s} {ms if ms is None else round(ms, 4)} ms" + ("" if line else "  " + r.stderr[-300:]))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
