#!/usr/bin/env python3
"""The training feed: windows per second of the HBM-resident slicer (beso_gather_windows)
against (a) what the HIP training step consumes and (b) the reference-style host feed -- a torch DataLoader over a
CPU slicer with the same semantics, pinned + prefetched to the device (beso_amd.data.prefetch.DevicePrefetcher).
    python tools/bench_feed.py [batch]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from beso_amd import synthetic as S  # noqa: E402
from beso_amd.data.prefetch import DevicePrefetcher  # noqa: E402
from beso_amd.data.trajectory_feed import DeviceTrajectoryFeed, window_table  # noqa: E402
from beso_amd.networks.scaler.scaler_class import Scaler  # noqa: E402
from _agent import build_agent  # noqa: E402


class HostSlicer(torch.utils.data.Dataset):
    """What the reference's loader does per item (trajectory_loader.py:160-197), on CPU tensors."""

    def __init__(self, obs, act, lengths, window, glen):
        self.obs, self.act, self.lengths, self.window, self.glen = torch.from_numpy(obs), torch.from_numpy(act), lengths, window, glen
        traj, start = window_table(lengths, window)
        self.slices = list(zip(traj.tolist(), start.tolist()))

    def __len__(self):
        return len(self.slices)

    def __getitem__(self, idx):
        i, start = self.slices[idx]
        end = start + self.window
        lo, hi = end, int(self.lengths[i]) - self.glen
        if lo < hi:
            g0 = np.random.randint(lo, hi)
            goal = self.obs[i][g0:g0 + self.glen]
        else:
            goal = torch.zeros(self.glen, self.obs.shape[2])
        return {"observation": self.obs[i][start:end], "action": self.act[i][start:end], "goal_observation": goal}


def main():
    dev = "cuda:0"
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    cfg = S.SHAPES["kitchen"]
    # relay-kitchen sized: 566 demonstrations padded to 409 steps (the reference's dataset is not shipped: synthetic)
    rng = np.random.default_rng(0)
    n, t_max = 566, 409
    lengths = rng.integers(160, t_max + 1, size=n).astype(np.int32)
    obs = rng.standard_normal((n, t_max, cfg.obs_dim)).astype(np.float32)
    act = rng.standard_normal((n, t_max, cfg.act_dim)).astype(np.float32)
    W, G = cfg.obs_seq_len, cfg.goal_seq_len
    feed = DeviceTrajectoryFeed(obs, act, lengths, W, B, dev, future_conditional=True, future_seq_len=G, seed=0)
    bytes_per_window = 2 * 4 * (W * (cfg.obs_dim + cfg.act_dim) + G * cfg.obs_dim)          # read + write

    # (1) the gather alone: HIP events around back-to-back launches on resident ids
    ids = torch.randint(0, feed.n_windows, (B,), device=dev)
    draws = torch.randint(0, 2 ** 62, (B,), device=dev)
    for _ in range(5):
        feed.gather(ids, draws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    reps = 200
    for _ in range(reps):
        feed.gather(ids, draws)
    e1.record()
    torch.cuda.synchronize()
    gather_ms = e0.elapsed_time(e1) / reps

    # (2) a whole epoch of dict batches (permutation + draws + gather + allocation), host clock
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nb = 0
    for batch in feed:
        nb += 1
    torch.cuda.synchronize()
    epoch_s = time.perf_counter() - t0

    # (3) train_step fed by it vs fed by one resident batch
    w = S.make_weights(cfg, seed=0, std=0.02)
    agent = build_agent(cfg, lambda: build_model(cfg, w, "bf16", dev), device=dev)
    agent.get_scaler(Scaler(obs[:8].reshape(-1, cfg.obs_dim), act[:8].reshape(-1, cfg.act_dim), True, dev))
    agent.set_bounds(agent.scaler)
    fixed = feed.gather(ids, draws)
    for _ in range(5):
        agent.train_step(fixed)

    def timed(batches, limit=60):
        torch.cuda.synchronize()
        t0, k = time.perf_counter(), 0
        for b in batches:
            agent.train_step(b)
            k += 1
            if k == limit:
                break
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k

    step_fixed = timed([fixed] * 60)
    step_feed = timed(feed)

    # (4) the reference-style host feed of the same windows: DataLoader workers + pinned prefetch
    host = HostSlicer(obs, act, lengths, W, G)
    rates = {}
    for workers in (0, 8):
        loader = torch.utils.data.DataLoader(host, batch_size=B, shuffle=True, num_workers=workers, pin_memory=False,
                                             persistent_workers=workers > 0)
        it = iter(DevicePrefetcher(loader, dev))
        next(it)
        t0, k = time.perf_counter(), 0
        for b in it:
            k += 1
            if k == 12:
                break
        torch.cuda.synchronize()
        rates[workers] = k * B / (time.perf_counter() - t0)
    print(json.dumps({
        "dataset": f"{n} trajectories x {t_max} steps (padded), obs {cfg.obs_dim}, act {cfg.act_dim}: {feed.n_windows} windows of {W} (+{G} goal)",
        "batch": B, "gather_kernel_us": 1e3 * gather_ms, "windows_per_s_gather": B / (gather_ms * 1e-3),
        "algorithmic_bytes_per_window": bytes_per_window, "gather_GBps": B * bytes_per_window / (gather_ms * 1e-3) / 1e9,
        "epoch_batches": nb, "windows_per_s_epoch_iteration": feed.n_windows / epoch_s,
        "train_step_ms_resident_batch": 1e3 * step_fixed, "train_step_ms_device_feed": 1e3 * step_feed,
        "windows_per_s_train_step_consumes": B / step_fixed,
        "windows_per_s_host_dataloader_0_workers": rates[0], "windows_per_s_host_dataloader_8_workers": rates[8]}))


if __name__ == "__main__":
    main()
