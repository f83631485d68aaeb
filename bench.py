#!/usr/bin/env python3
"""Headline benchmark: denoising-steps/sec of the score-GPT forward at the Franka-kitchen shape.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE denoising step = one GCDenoiser.forward (score-GPT forward + Karras preconditioning)
over one batch of synthetic kitchen inputs: BASELINE.json configs[1] (kitchen, B=4096 per GPU, bf16
MFMA, inputs resident in HBM).  Samples are independent, so N GPUs run N shards of the job with no
collective in the loop (weak scaling: 4096 samples per GPU per step); value = all steps of all
ranks / max-over-ranks time.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches this script under
`torch.distributed.run` (one process per GPU, RCCL); the ranks verify the group with an all-reduce of their rank
numbers before anything is timed.  A world size that disagrees with --gpus, or fewer visible GPUs than N, is an error.

Rank 0 prints ONE JSON line with the driver's fields plus
  "roofline":     the dominant kernel's achieved TFLOP/s (algorithmic FLOPs / HIP-event time on the
                  launch stream) against the dense bf16 MFMA peak; `traffic` = HBM bytes per launch from two
                  rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of this very script, run as child processes;
  "parity_mode":  the same forward in the split-bf16 (BF16X3) instance of the same kernel -- the mode that meets the
                  north-star 1e-4 tolerance -- timed beside the bf16 headline;
  "fp16_mode":    the same forward through the fp16-operand build of the same kernel (BESO_PREC_FP16), with the deviation
                  of both 16-bit modes from the parity mode's output;
  "cold_ms_per_step" / "long_run": the K steps timed straight behind the W warm-up steps (before the clock-settle loop),
                  and >= 200 steps of the settled loop -- `value` itself is exactly K steps behind the settle loop;
  "cpu_baseline": the reference's CPU path (ATen restatement in oracle/beso_oracle_torch.py) timed on this box's
                  host cores on a bounded sample of the same workload.

`--workload train` times BASELINE configs[2]'s per-GPU share instead: BesoAgent.train_step (score-matching loss,
backward, AdamW, EMA; kitchen dropouts and cond_mask_prob = 0.1) at 1024 samples per GPU, gradients all-reduced over
the ranks (C1, `--c1-overlap 0|1`).
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with `hipIpcGetMemHandle: invalid argument`
# otherwise); the launcher normally exports it -- set it before the HIP runtime comes up in case it did not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "bf16x3": 2500.0, "fp32": 157.3}       # dense MFMA peaks (MI355X_MICROARCH.md)


def build_model(cfg, w, precision, dev, attn_pdrop=0.0, resid_pdrop=0.0, goal_drop=0.0, train=False):
    from beso_amd.agents.diffusion_agents.k_diffusion.score_gpts import DiffusionGPT
    from beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser
    inner = DiffusionGPT(state_dim=cfg.obs_dim, device=dev, goal_conditioned=cfg.goal_conditioned,
                         action_dim=cfg.act_dim, embed_dim=cfg.embed_dim, embed_pdrob=0, attn_pdrop=attn_pdrop,
                         resid_pdrop=resid_pdrop, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=cfg.goal_seq_len,
                         obs_seq_len=cfg.obs_seq_len, sigma_vocab_size=3, time_embedding_fn=None, goal_drop=goal_drop,
                         linear_output=cfg.linear_output, precision=precision)
    m = GCDenoiser(inner, sigma_data=cfg.sigma_data)
    sd = m.state_dict()
    sd.update({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    m.load_state_dict(sd)
    m = m.to(dev)
    return m.train() if train else m.eval()


# ------------------------------------------------------------------------------------------------
# launcher: --gpus N without a process group around us -> one rank per GPU under torch.distributed.run
# ------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def maybe_relaunch(args):
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch one rank per GPU "
                     f"(torch.distributed.run --nproc-per-node {args.gpus}) or drop WORLD_SIZE")
        return
    if args.gpus <= 1:
        return
    if args.dry_run_backend is None:
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} requested but {n_dev} GPU(s) are visible to this process: "
                     f"refusing to run fewer ranks than asked for")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    sys.exit(subprocess.call(cmd))


def init_ranks(args):
    """-> (world, rank, local_rank, ranks_verified).  The process group is proven by an all-reduce of the rank numbers."""
    from beso_amd import distributed as bdist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    verified = 1
    if world > 1:
        backend = args.dry_run_backend or "nccl"
        bdist.init_from_env(backend)
        import torch.distributed as dist
        assert dist.get_world_size() == args.gpus == world, (dist.get_world_size(), args.gpus, world)
        dev = "cpu" if backend == "gloo" else f"cuda:{local_rank}"
        t = torch.tensor([float(rank), 1.0], device=dev)
        dist.all_reduce(t)
        if t[0].item() != world * (world - 1) / 2 or t[1].item() != world:
            raise SystemExit(f"bench.py: the {backend} group does not contain {world} distinct ranks (sum {t.tolist()})")
        verified = int(t[1].item())
    return world, rank, local_rank, verified


# ------------------------------------------------------------------------------------------------
# CPU baseline: the reference's CPU path (ATen ops, fp32) on this box's host cores, bounded
# ------------------------------------------------------------------------------------------------
def cpu_baseline(cfg, w, batch, budget_s=12.0):
    """oracle/beso_oracle_torch.py -- the ATen restatement of GCDenoiser.forward / sample_ddim that the CPU suite pins to
    the reference's vectors -- timed on the host cores.  The ONLY place where bench.py touches oracle/: it is the thing
    timed here, on the same synthetic workload, never a part of the GPU path."""
    from oracle import beso_oracle as O
    from oracle import beso_oracle_torch as OT
    from beso_amd import synthetic as S
    ocfg = O.ScoreGPTConfig(**cfg.as_dict())
    W = OT.to_torch(w)
    cores = os.cpu_count() or 1
    # ATen's intra-op pool stops scaling on these [B*11, 360] x [360, 1440] problems early: measured on the GPU box's
    # host (2 x EPYC 9575F, 256 hardware threads; tests/cpu_scan.py) 8 / 16 / 32 / 64 / 128 threads give 1465 / 1709 /
    # 1734 / 973 / 430 samples/s at B = 4096 and 2408 / 3629 / 2071 / 908 / 205 at B = 64 -- 16 is the best setting
    threads = min(cores, 16)
    torch.set_num_threads(threads)
    s, g, a = (torch.from_numpy(v) for v in S.make_inputs(cfg, batch, seed=0))
    sig = torch.full((batch,), 0.3)
    OT.denoise(W, ocfg, s[:64], a[:64], g[:64], sig[:64])                     # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    n = 0
    while True:
        OT.denoise(W, ocfg, s, a, g, sig)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 20:
            break
    steps_per_s = n / el
    # BASELINE configs[0]: kitchen, B = 64, 10 DDIM steps, reference on CPU
    s64, g64, x64 = (torch.from_numpy(v) for v in S.make_inputs(cfg, 64, seed=1))
    sigmas = torch.from_numpy(O.get_sigmas_exponential(10, 0.005, 1.0))
    OT.sample_ddim(W, ocfg, s64, x64, g64, sigmas[-3:])
    t1 = time.perf_counter()
    reps = 0
    while True:
        OT.sample_ddim(W, ocfg, s64, x64, g64, sigmas)
        reps += 1
        el1 = time.perf_counter() - t1
        if el1 > 5.0 or reps >= 10:
            break
    cfg1_ms = 1e3 * el1 / reps
    multi = cpu_baseline_multiprocess(cfg, batch, cores)
    return {"value": steps_per_s, "unit": f"denoise-steps/s (B={batch} per step)", "cores": int(threads),
            "kind": "port",
            "sample": f"{n} GCDenoiser.forward calls over B={batch} kitchen samples, sigma=0.3 ({el:.1f} s of ATen fp32 "
                      f"with {threads} threads on a {cores}-core host)",
            "samples_per_s": steps_per_s * batch,
            "host_cores": cores,
            "all_cores": multi,
            "config0_b64_ddim10": {"ms": cfg1_ms, "denoise_steps_per_s": 10.0 / (cfg1_ms * 1e-3),
                                   "sample_steps_per_s": 640.0 / (cfg1_ms * 1e-3), "threads": int(threads),
                                   "what": "BASELINE configs[0]: kitchen, B=64, 10 DDIM steps, fp32 on CPU"}}


def physical_core_sets(threads):
    """Disjoint sets of `threads` PHYSICAL cores (one hardware thread per core, all of one package) for pinned workers:
    [[cpu, ...], ...].  Round 3's all-core figure (16 unpinned processes x 16 threads on 256 hardware threads) came out
    BELOW one 16-thread process -- oversubscribed SMT siblings and threads wandering across the sockets, not a baseline."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return []
    by_pkg, seen = {}, set()
    for c in allowed:
        base = f"/sys/devices/system/cpu/cpu{c}/topology/"
        try:
            sib = open(base + "thread_siblings_list").read().strip()
            pkg = int(open(base + "physical_package_id").read())
        except (OSError, ValueError):
            sib, pkg = str(c), 0
        if sib in seen:
            continue                                   # an SMT sibling of a core that is already taken
        seen.add(sib)
        by_pkg.setdefault(pkg, []).append(c)
    sets = []
    for pkg in sorted(by_pkg):
        cs = by_pkg[pkg]
        sets += [cs[i:i + threads] for i in range(0, len(cs) - threads + 1, threads)]
    return sets


def cpu_baseline_multiprocess(cfg, batch, cores, run_s=8.0):
    """The same ATen forward on ALL physical host cores: samples are independent, so P processes x 16 threads each take a
    disjoint shard of the batch (ATen's intra-op pool stops scaling at ~16 threads on these GEMMs; processes do not share
    it), every process PINNED to its own 16 physical cores of one socket (os.sched_setaffinity).  Every worker warms up,
    waits for a common start time, runs its shard repeatedly for `run_s` seconds and reports (samples processed,
    elapsed); the figure is sum(samples) / max(elapsed)."""
    threads = 16
    sets = physical_core_sets(threads)[:16]
    procs = len(sets)
    if procs <= 1:
        return None
    shard = max(16, batch // procs)
    t_start = time.time() + 12.0                        # (imports + warm-up of the workers)
    from beso_amd import synthetic as S
    name = next(k for k, v in S.SHAPES.items() if v == cfg)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", name,
           str(shard), str(threads), repr(t_start), repr(run_s)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    kids = [subprocess.Popen(cmd + [str(i), ",".join(map(str, sets[i]))], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             text=True, env=env) for i in range(procs)]
    res = []
    for k in kids:
        try:
            out, _ = k.communicate(timeout=run_s + 60.0)
            line = [ln for ln in out.splitlines() if ln.startswith("{")]
            if line:
                res.append(json.loads(line[-1]))
        except Exception:      # noqa: BLE001
            k.kill()
    if len(res) != procs:
        return {"error": f"{len(res)} of {procs} workers reported"}
    samples = sum(r["samples"] for r in res)
    el = max(r["elapsed"] for r in res)
    return {"samples_per_s": samples / el, "denoise_steps_per_s_at_this_batch": samples / el / batch, "processes": procs,
            "threads_per_process": threads, "cores": procs * threads, "shard": shard, "run_s": el, "pinned": True,
            "what": f"{procs} processes x {threads} ATen threads, each pinned to its own {threads} physical cores of one socket, "
                    f"each the fp32 forward over its own {shard}-sample shard, common start, {run_s:.0f} s"}


def cpu_worker(argv):
    name, shard, threads, t_start, run_s, idx = argv[0], int(argv[1]), int(argv[2]), float(argv[3]), float(argv[4]), int(argv[5])
    if len(argv) > 6 and argv[6]:
        try:
            os.sched_setaffinity(0, {int(c) for c in argv[6].split(",")})      # before the thread pools exist
        except (AttributeError, OSError, ValueError):
            pass
    from oracle import beso_oracle as O
    from oracle import beso_oracle_torch as OT
    from beso_amd import synthetic as S
    torch.set_num_threads(threads)
    cfg = S.SHAPES[name]
    ocfg = O.ScoreGPTConfig(**cfg.as_dict())
    W = OT.to_torch(S.make_weights(cfg, seed=0, std=0.02))
    s, g, a = (torch.from_numpy(v) for v in S.make_inputs(cfg, shard, seed=100 + idx))
    sig = torch.full((shard,), 0.3)
    OT.denoise(W, ocfg, s[:16], a[:16], g[:16], sig[:16])
    while time.time() < t_start:
        time.sleep(0.005)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < run_s:
        OT.denoise(W, ocfg, s, a, g, sig)
        n += 1
    print(json.dumps({"samples": n * shard, "elapsed": time.perf_counter() - t0}))


# ------------------------------------------------------------------------------------------------
# HBM traffic of the dominant kernel: two rocprofv3 PMC passes over a short child run of this script
# ------------------------------------------------------------------------------------------------
def measure_traffic(args, kernel_substr="layers_kernel"):
    """FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots: MI355X_MICROARCH.md, rocprofv3 PMC slots), so each gets
    its own run with --kernel-trace only.  Both are reported in KiB; the read side is doubled (gfx950: FETCH_SIZE counts
    128-B requests as 64 B).  Returns (bytes per launch or None, detail dict)."""
    if os.environ.get("BESO_BENCH_TRAFFIC", "1") == "0" or shutil.which("rocprofv3") is None:
        return None, {"source": "not measured (rocprofv3 unavailable or BESO_BENCH_TRAFFIC=0)"}
    import csv
    vals = {}
    tmp = tempfile.mkdtemp(prefix="beso_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", BESO_BENCH_TRAFFIC="0")
    child = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "4", "--warmup", "2", "--batch", str(args.batch),
             "--precision", args.precision, "--config", args.config, "--no-cpu-baseline", "--no-parity-line", "--no-other-configs", "--settle-ms", "0"]
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "t", "--", *child]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            if r.returncode != 0:
                return None, {"source": f"rocprofv3 pass {counter} failed (rc {r.returncode})"}
            tot, cnt = 0.0, 0
            for dp, _, files in os.walk(out):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        for row in csv.DictReader(open(os.path.join(dp, f))):
                            if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter:
                                tot += float(row["Counter_Value"])
                                cnt += 1
            if cnt == 0:
                return None, {"source": f"no {kernel_substr} rows in the {counter} pass"}
            vals[counter] = (tot / cnt, cnt)
    except Exception as e:        # noqa: BLE001   (a profiler hiccup must not cost the bench line)
        return None, {"source": f"traffic measurement failed: {type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch_kib, write_kib = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
    total = (2.0 * fetch_kib + write_kib) * 1024.0
    return total, {"source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) "
                             "over child runs of bench.py, averaged over the launches of " + kernel_substr,
                   "FETCH_SIZE_KiB_avg": fetch_kib, "WRITE_SIZE_KiB_avg": write_kib, "launches_sampled": vals["FETCH_SIZE"][1],
                   "formula": "2 * FETCH_SIZE (gfx950 correction) + WRITE_SIZE"}


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def _sync():
    if torch.cuda.is_available():       # (the gloo dry runs of the N > 1 paths have no device to wait for)
        torch.cuda.synchronize()


def settle(step, ms):
    """Untimed: run the step until the clocks have left their idle state (a cold MI355X takes ~100 ms of load to reach
    its sustained clock; with 5 warm-up steps of 0.9 ms the first timed steps run ~4 % slow)."""
    if ms <= 0:
        return 0
    n = 0
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(20):
            step()
        _sync()
        n += 20
    return n


def timed(step, steps, world):
    _sync()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    _sync()
    if world > 1:
        torch.distributed.barrier()
    return time.perf_counter() - t0, out


def run_forward(args, world, rank, dev):
    from beso_amd import synthetic as S
    cfg = S.SHAPES[args.config]
    w = S.make_weights(cfg, seed=0, std=0.02)
    model = build_model(cfg, w, args.precision, dev)
    inner = model.inner_model
    B = args.batch
    s_np, g_np, a_np = S.make_inputs(cfg, B, seed=1000 + rank)          # every rank its own shard of the job
    state, goal, action = (torch.from_numpy(v).to(dev) for v in (s_np, g_np, a_np))
    sigma = torch.full((B,), 0.3, device=dev)
    rt = inner.runtime(cfg.sigma_data)

    def step():
        # one step = the drop-in module's call, as a workspace makes it: GCDenoiser.forward(state, action, goal, sigma)
        # (score_wrappers.py:81-96) -- incl. the packed-weight cache check of every call, not a pre-fetched image
        return model(state, action, goal, sigma)

    with torch.no_grad():
        for _ in range(args.warmup):
            out = step()
        torch.cuda.synchronize()
        # the driver's protocol taken literally: K steps straight behind the W warm-up steps (a cold MI355X is still ramping)
        cold_elapsed, _ = timed(step, args.steps, world)
        settled = settle(step, args.settle_ms)
        # which launch site dominates?  time each once, then instrument the dominant one
        sites = ["fused_layer", "gemm_fc1", "gemm_fc2", "gemm_qkv", "gemm_proj", "attention", "layernorm"]
        site_ms = {}
        for site in sites:
            rt.profile_enable(site)
            step()
            torch.cuda.synchronize()
            ms, n = rt.profile_read()
            if n:
                site_ms[site] = (ms, n)
        rt.profile_enable("off")
        dominant = args.site or max(site_ms, key=lambda k: site_ms[k][0])
        rt.profile_enable(dominant)
        elapsed, out = timed(step, args.steps, world)
        kern_ms, kern_n = rt.profile_read()
        rt.profile_enable("off")
        assert torch.isfinite(out).all()
        # ... and a region long enough to carry a sub-percent margin (>= 200 steps of the same settled loop)
        long_steps = max(200, args.steps)
        long_elapsed, _ = timed(step, long_steps, world)
        # the parity mode of the same kernel, timed beside the headline (rank 0, N = 1: it is not part of `value`)
        parity = fp16 = None
        if world == 1 and not args.no_parity_line and args.precision == "bf16" and args.config in ("kitchen", "block_push"):
            mx = build_model(cfg, w, "bf16x3", dev)
            ix = mx.inner_model
            rtx = ix.runtime(cfg.sigma_data)
            stepx = lambda: mx(state, action, goal, sigma)      # noqa: E731
            for _ in range(3):
                outx = stepx()
            rtx.profile_enable("fused_layer")
            nx = max(3, min(args.steps, 10))
            elx, outx = timed(stepx, nx, 1)
            kx_ms, kx_n = rtx.profile_read()
            rtx.profile_enable("off")
            dev_rel = float((outx - out).abs().max() / out.abs().max())
            parity = {"dtype": "bf16x3", "ms_per_step": 1e3 * elx / nx, "kernel_avg_launch_ms": kx_ms / max(kx_n, 1),
                      "launches": kx_n, "max_rel_dev_of_bf16_from_this_mode": dev_rel,
                      "what": "same GCDenoiser.forward through the split-bf16 instance of layers_kernel (3 MFMAs per operand "
                              "pair, exact GELU, fp32 attention core; four samples in three token tiles per workgroup): "
                              "1e-4-class parity with the fp32 reference (tests/test_gpu_parity.py: 5e-6 .. 3e-5).  "
                              "Ceiling of ANY 1e-4 mode of this kernel: a 1e-4 result needs >= 16 significand bits per operand, "
                              "i.e. three bf16 MFMAs per product (two leave 2^-12 per element), so its algorithmic-FLOP fraction "
                              "is at most a third of the bf16 instance's (forward_frac_ceiling below); what this line shows beyond "
                              "that factor is the mode's own overhead (both weight images streamed, exact GELU, fp32 core)"}
            # the fp16-operand build of the same kernel, the same way; its deviation from the parity mode beside bf16's
            mh = build_model(cfg, w, "fp16", dev)
            ih = mh.inner_model
            rth = ih.runtime(cfg.sigma_data)
            steph = lambda: mh(state, action, goal, sigma)      # noqa: E731
            for _ in range(20):
                outh = steph()
            rth.profile_enable("fused_layer")
            nh = max(50, args.steps)
            elh, outh = timed(steph, nh, 1)
            kh_ms, kh_n = rth.profile_read()
            rth.profile_enable("off")
            fp16 = {"dtype": "fp16", "ms_per_step": 1e3 * elh / nh, "kernel_avg_launch_ms": kh_ms / max(kh_n, 1), "launches": kh_n,
                    "max_rel_dev_from_parity_mode": float((outh - outx).abs().max() / outx.abs().max()),
                    "bf16_max_rel_dev_from_parity_mode": dev_rel,
                    "what": "same GCDenoiser.forward through the fp16-operand build of layers_kernel (BESO_PREC_FP16: "
                            "v_mfma_f32_16x16x32_f16, the bf16 MFMA rate, 11-bit significands); vs the reference's vectors "
                            "1.3e-3 .. 1.8e-3 where bf16 measures 7e-3 .. 1.2e-2 (tests/test_gpu_parity.py)"}
    if world > 1:
        tmax = torch.tensor([elapsed, cold_elapsed, long_elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed, cold_elapsed, long_elapsed = (float(v) for v in tmax.tolist())
    if rank != 0:
        return None
    F = cfg.flops_per_sample()
    D, T, t = cfg.embed_dim, cfg.block_size, cfg.obs_seq_len
    M = B * T
    # algorithmic FLOPs of ONE launch of each launch site: the REFERENCE's work
    layers_flops = (24.0 * T * D * D + 4.0 * T * T * D) * B * cfg.n_layers
    site_flops = {"gemm_qkv": 2.0 * M * 3 * D * D, "gemm_proj": 2.0 * M * D * D, "gemm_fc1": 2.0 * M * 4 * D * D,
                  "gemm_fc2": 2.0 * M * 4 * D * D, "attention": 4.0 * B * T * T * D,
                  "fused_layer": layers_flops, "forward": float(F) * B}
    executed = dict(site_flops)
    if "fused_layer" in site_ms and site_ms["fused_layer"][1] > 1:
        # shapes without the fused attention phase launch the fused MLP block once per layer
        site_flops["fused_layer"] = executed["fused_layer"] = 16.0 * T * D * D * B
        kernel_symbol = "beso::mlp_block_kernel (LN2+FC1+GELU+FC2+residual, one launch per layer)"
    else:
        # the fused kernel runs ALL layers in one launch; in the LAST layer only the action tokens go through the
        # out-projection and the MLP (nothing else reaches the head): 18 D^2 per skipped token
        executed["fused_layer"] = layers_flops - 18.0 * (T - t) * D * D * B
        kernel_symbol = "beso::layers_kernel<3,12> (all transformer layers, one launch)"
    steps_per_s = world * args.steps / elapsed
    fwd_tflops = B * F * args.steps / elapsed / 1e12          # per GPU
    avg_ms = kern_ms / max(kern_n, 1)
    ach = site_flops.get(dominant, 0.0) / (avg_ms * 1e-3) / 1e12 if kern_n else 0.0
    ach_x = executed.get(dominant, 0.0) / (avg_ms * 1e-3) / 1e12 if kern_n else 0.0
    peak = PEAK_TFLOPS[args.precision]
    traffic, traffic_detail = (None, {"source": "not measured (N > 1 or --no-traffic)"})
    if world == 1 and not args.no_traffic and dominant == "fused_layer":
        traffic, traffic_detail = measure_traffic(args)
    if parity is not None:
        parity["forward_frac_of_bf16_mfma_peak"] = B * F / (parity["ms_per_step"] * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"]
        parity["executed_mfma_flops_over_algorithmic"] = 3.0
        parity["forward_frac_ceiling"] = B * F / (1e3 * elapsed / args.steps * 1e-3) / 1e12 / PEAK_TFLOPS["bf16"] / 3.0
    result = {
        "metric": "denoising-steps/sec (score-GPT fwd) at kitchen obs-dim",
        "value": steps_per_s, "unit": f"denoise-steps/s (one step = GCDenoiser.forward over B={B} samples per GPU)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[1]: Franka {args.config} score-GPT fwd "
                               f"(GCDenoiser.forward), batch={B} synthetic obs/goal per GPU, 1 NFE per step",
                   "obs_dim": cfg.obs_dim, "act_dim": cfg.act_dim, "window": cfg.obs_seq_len,
                   "goal_seq_len": cfg.goal_seq_len, "embed_dim": D, "n_layers": cfg.n_layers,
                   "n_heads": cfg.n_heads, "tokens_per_sample": T, "batch_per_gpu": B,
                   "parallelism": f"batch-sharded x{world}, no collective in the loop",
                   "weights": "seeded N(0,0.02) recipe (no trained checkpoints shipped)", "sigma": 0.3},
        "untimed_settle_steps": settled,
        "cold_ms_per_step": 1e3 * cold_elapsed / args.steps,
        "long_run": {"steps": long_steps, "ms_per_step": 1e3 * long_elapsed / long_steps,
                     "frac_of_mfma_peak": B * F / (long_elapsed / long_steps) / 1e12 / peak},
        "sample_nfe_per_s": steps_per_s * B,
        "flops_per_sample_forward": F,
        "forward_tflops_per_gpu": fwd_tflops,
        "forward_frac_of_mfma_peak": fwd_tflops / peak,
        "roofline": {"bound": "mfma", "kernel": dominant, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                     "frac": ach / peak, "traffic": traffic,
                     "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE)", "traffic_detail": traffic_detail,
                     "kernel_symbol": kernel_symbol if dominant == "fused_layer" else dominant, "launches": kern_n,
                     "avg_launch_ms": avg_ms, "flops_per_launch": site_flops.get(dominant, 0.0),
                     "executed_flops_per_launch": executed.get(dominant, 0.0), "achieved_executed": ach_x,
                     "frac_executed": ach_x / peak,
                     "site_ms_one_forward": {k: v[0] for k, v in site_ms.items()}},
    }
    if parity is not None:
        result["parity_mode"] = parity
    if fp16 is not None:
        fp16["forward_frac_of_fp16_mfma_peak"] = B * F / (fp16["ms_per_step"] * 1e-3) / 1e12 / PEAK_TFLOPS["fp16"]
        result["fp16_mode"] = fp16
    if not args.no_cpu_baseline and world == 1:          # rank 0 at N=1 only (the other ranks would idle at the barrier)
        result["cpu_baseline"] = cpu_baseline(cfg, w, B)
    if world == 1 and not args.no_other_configs and args.config == "kitchen" and args.precision == "bf16":
        result["other_configs"] = other_configs(args, dev)
    return result


def other_configs(args, dev):
    """The other BASELINE.json configurations, measured in the same run (N = 1 only, after the timed region of the headline)
    through the public sampler API / BesoAgent.train_step, a few calls each -- so that the record of this command also carries
    configs 0, 2, 3 and 4.  tools/bench_configs.py / tools/bench_train.py are the longer stand-alone forms."""
    import copy
    from beso_amd import synthetic as S
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    out = []

    def sampler_run(name, shape, B, sampler, n_steps, smin, smax, lam, reps, precision="bf16"):
        cfg = S.SHAPES[shape]
        model = build_model(cfg, S.make_weights(cfg, seed=0, std=0.02), precision, dev)
        call = model if lam is None else ClassifierFreeSampleModel(model, lam)
        s, g, a = (torch.from_numpy(v).to(dev) for v in S.make_inputs(cfg, B, seed=1))
        x_t = torch.randn_like(a) * smax
        sigmas = ks.get_sigmas_exponential(n_steps, smin, smax)
        fn = {"ddim": ks.sample_ddim, "euler": ks.sample_euler, "heun": ks.sample_heun}[sampler]
        nfe = n_steps if sampler != "heun" else 2 * n_steps - 1
        per_nfe = 2 if (lam is not None and lam not in (0.0, 1.0)) else 1
        with torch.no_grad():
            fn(call, s, x_t, g, sigmas, disable=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                res = fn(call, s, x_t, g, sigmas, disable=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
        assert torch.isfinite(res).all()
        flops = cfg.flops_per_sample() * B * nfe * per_nfe
        out.append({"config": name, "batch": B, "sampler": sampler, "steps": n_steps, "nfe": nfe, "cond_lambda": lam, "dtype": precision,
                    "calls_timed": reps, "ms_per_call": 1e3 * dt, "denoise_steps_per_s": n_steps / dt,
                    "sample_nfe_per_s": B * nfe / dt, "tflops": flops / dt / 1e12, "frac_of_bf16_mfma_peak": flops / dt / 1e12 / PEAK_TFLOPS["bf16"]})

    sampler_run("configs[0]: kitchen B=64 DDIM-10", "kitchen", 64, "ddim", 10, 0.005, 1.0, None, 20)
    sampler_run("configs[3]: block-push B=2048 Heun-50 x CFG lambda=2", "block_push", 2048, "heun", 50, 0.05, 1.0, 2.0, 3)
    sampler_run("configs[4]: long-horizon (D=512, 67 tokens) B=256 per GPU Euler-100", "long_horizon", 256, "euler", 100, 0.005, 1.0, None, 3)
    # (a sample per workgroup: 256 is exactly one workgroup per CU; 384 = 1.5 rounds and 1024 = 4 rounds beside it)
    sampler_run("configs[4] at B=384 per GPU", "long_horizon", 384, "euler", 100, 0.005, 1.0, None, 2)
    sampler_run("configs[4] at B=1024 per GPU", "long_horizon", 1024, "euler", 100, 0.005, 1.0, None, 1)
    # the other precisions of the one-launch kernel on the sampler configurations (fp16 operands; the 1e-4 mode)
    sampler_run("configs[3] in fp16", "block_push", 2048, "heun", 50, 0.05, 1.0, 2.0, 2, "fp16")
    sampler_run("configs[4] in fp16", "long_horizon", 256, "euler", 100, 0.005, 1.0, None, 2, "fp16")
    sampler_run("configs[3] in bf16x3 (1e-4 mode)", "block_push", 2048, "heun", 50, 0.05, 1.0, 2.0, 1, "bf16x3")
    sampler_run("configs[4] in bf16x3 (1e-4 mode: split-bf16 block kernels + split-bf16 MFMA attention)", "long_horizon", 256, "euler", 100, 0.005, 1.0, None, 1, "bf16x3")
    for label, shape, B in (("configs[2] per-GPU share: kitchen BesoAgent.train_step, 1024 samples", "kitchen", 1024),
                            ("configs[2] whole on one GPU: kitchen BesoAgent.train_step, 8192 samples", "kitchen", 8192),
                            ("block-push BesoAgent.train_step (resid_pdrop 0.05), 1024 samples", "block_push", 1024)):
        targs = copy.copy(args)
        targs.workload, targs.batch, targs.steps, targs.warmup, targs.settle_ms, targs.config = "train", B, 20, 3, 100.0, shape
        tr = run_train(targs, 1, 0, dev)
        out.append({"config": label, "dtype": "bf16",
                    "steps_timed": targs.steps, "ms_per_step": tr["ms_per_step"], "samples_per_s": tr["samples_per_s"],
                    "tflops": tr["roofline"]["achieved"], "frac_of_bf16_mfma_peak": tr["roofline"]["frac"], "loss": tr["loss"]})
    out.append(small_batches(dev))
    return out


def small_batches(dev):
    """The rollout end of the batch-size range (the reference's workspaces call predict() with B = 1): one GCDenoiser.forward and
    a 3-step sampler call at 1 / 16 samples through the library's own choice (bf16, kitchen: the chip-wide small-batch path,
    small.hip) next to the one-launch kernel's latency instance (BESO_PLAN_FUSED), and the same forward in fp32 (small-batch path
    against the per-op kernels)."""
    from beso_amd import _lib, synthetic as S
    from beso_amd.runtime import set_plan
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    cfg = S.SHAPES["kitchen"]
    w = S.make_weights(cfg, seed=0, std=0.02)
    rec = {"config": "small batches (kitchen): one forward / a 3-step DDIM call, microseconds", "rows": []}

    def us(fn, n=200, warm=20):
        with torch.no_grad():
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    try:
        for prec, other, hint in (("bf16", "one_launch_kernel", _lib.PLAN_FUSED), ("fp32", "per_op_kernels", _lib.PLAN_PER_OP)):
            model = build_model(cfg, w, prec, dev)
            sig3 = ks.get_sigmas_exponential(3, 0.005, 1.0)
            for B in (1, 16):
                s, g, a = (torch.from_numpy(v).to(dev) for v in S.make_inputs(cfg, B, seed=1))
                sg = torch.full((B,), 0.3, device=dev)
                row = {"dtype": prec, "batch": B}
                for name, h in (("library", 0), (other, hint)):
                    set_plan(forward=h)
                    row[f"forward_us_{name}"] = us(lambda: model(s, a, g, sg))
                    row[f"ddim3_us_{name}"] = us(lambda: ks.sample_ddim(model, s, a, g, sig3, disable=True), n=50, warm=5)
                rec["rows"].append(row)
    finally:
        set_plan(forward=0)
    return rec



def run_train(args, world, rank, dev):
    """BASELINE configs[2] per-GPU share: kitchen train_step, 1024 samples per GPU, shipped dropouts, cond_mask_prob 0.1."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from _agent import build_agent
    from beso_amd import distributed as bdist
    from beso_amd import synthetic as S
    from beso_amd.networks.scaler.scaler_class import Scaler
    os.environ["BESO_AMD_C1_OVERLAP"] = str(args.c1_overlap)
    cfg = S.SHAPES[args.config]
    w = S.make_weights(cfg, seed=0, std=0.02)
    B = args.batch
    attn_p, resid_p = {"kitchen": (0.3, 0.0), "block_push": (0.05, 0.05)}.get(args.config, (0.0, 0.0))
    agent = build_agent(cfg, lambda: build_model(cfg, w, args.precision, dev, attn_p, resid_p, 0.1, train=True), device=dev)
    rng = np.random.default_rng(0)
    agent.get_scaler(Scaler(rng.standard_normal((256, cfg.obs_dim)).astype(np.float32),
                            rng.standard_normal((256, cfg.act_dim)).astype(np.float32), True, dev))
    agent.set_bounds(agent.scaler)
    agent._sync_replicas()                                  # C2: replicas start identical, EMA re-seeded
    torch.manual_seed(1234 + rank)                          # every rank its own samples, noise, sigma and masks
    batch = {"observation": torch.randn(B, cfg.obs_seq_len, cfg.obs_dim, device=dev),
             "action": torch.randn(B, cfg.obs_seq_len, cfg.act_dim, device=dev),
             "goal_observation": torch.randn(B, cfg.goal_seq_len, cfg.obs_dim, device=dev)}
    step = lambda: agent.train_step(batch)                  # noqa: E731
    for _ in range(max(args.warmup, 2)):
        loss = step()
    settle(step, args.settle_ms)
    elapsed, loss = timed(step, args.steps, world)
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if getattr(args, "keep_agent", None) is not None:
        args.keep_agent.append(agent)                       # (tests: the replicas' parameters after the run)
    if rank != 0:
        return None
    assert np.isfinite(loss) and (getattr(agent, "_hip_step", None) is not None or str(dev) == "cpu")
    flops = 3.0 * cfg.flops_per_sample() * B                # forward + data gradients + weight gradients, per GPU
    ms = 1e3 * elapsed / args.steps
    ach = flops / (ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[args.precision]
    return {
        "metric": "train-steps/sec (score-matching train_step) at kitchen obs-dim",
        "value": args.steps / elapsed, "unit": f"train_steps/s of the whole job (global batch {B * world}: {B} samples per GPU per step)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[2] per-GPU share: Franka {args.config} BesoAgent.train_step "
                               f"(GCDenoiser.loss fwd+bwd, AdamW, EMA), {B} samples per GPU, global batch {B * world}",
                   "attn_pdrop": attn_p, "resid_pdrop": resid_p, "cond_mask_prob": 0.1,
                   "parallelism": f"dp{world}: all-reduce of {9381249 if args.config == 'kitchen' else 'all'} fp32 gradients "
                                  f"(C1, overlap={'on' if args.c1_overlap else 'off'})",
                   "optimizer": type(agent.optimizer).__name__ + " + EMA in one launch"},
        "samples_per_s": B * world * args.steps / elapsed,
        "loss": loss,
        "roofline": {"bound": "mfma", "kernel": "train_step (all launches of one step)", "achieved": ach, "peak": peak,
                     "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                     "flops_per_launch": flops, "avg_launch_ms": ms,
                     "note": "3 x forward FLOPs x samples per GPU over the whole step time (optimizer, EMA, C1 included)"},
    }


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU per step (forward: 4096, train: 1024)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32", "bf16x3"])
    ap.add_argument("--config", default="kitchen", choices=["kitchen", "block_push", "long_horizon"])
    ap.add_argument("--workload", default="forward", choices=["forward", "train"])
    ap.add_argument("--c1-overlap", type=int, default=1, choices=[0, 1], help="train workload, N > 1: overlapped gradient all-reduce")
    ap.add_argument("--settle-ms", type=float, default=300.0, help="untimed load before the timed region (clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-line", action="store_true")
    ap.add_argument("--no-traffic", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE configs 0, 2, 3, 4 behind the headline")
    ap.add_argument("--site", default=None, help="launch site timed for the roofline object")
    ap.add_argument("--dry-run-backend", default=None, choices=["gloo"],
                    help="launcher self-test without GPUs: spawn the ranks, verify the group over this backend, print n_gpus")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 4096 if args.workload == "forward" else 1024
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    maybe_relaunch(args)
    world, rank, local_rank, verified = init_ranks(args)
    if args.dry_run_backend is not None:
        if rank == 0:
            print(json.dumps({"dry_run": True, "backend": args.dry_run_backend, "n_gpus": world, "ranks_verified": verified}))
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible (the score-denoising path has no CPU implementation to time)")
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(dev)
    from beso_amd import distributed as bdist
    numa = bdist.pin_to_gpu_numa_node(local_rank) if world > 1 else None       # one rank per GPU: host threads beside it
    result = run_forward(args, world, rank, dev) if args.workload == "forward" else run_train(args, world, rank, dev)
    if rank == 0:
        result["ranks_verified"] = verified
        result["numa_node_of_rank0"] = numa
        print(json.dumps(result))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
