#!/usr/bin/env python3
"""Where a training step's WALL time goes: host enqueue time against GPU time (is the step launch bound?).
    python tools/train_host_profile.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from beso_amd import synthetic as O  # noqa: E402
from _agent import build_agent  # noqa: E402
from beso_amd.networks.scaler.scaler_class import Scaler  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = "cuda:0"
    cfg = O.SHAPES["kitchen"]
    w = O.make_weights(cfg, seed=0, std=0.02)

    def model():
        m = build_model(cfg, w, "bf16", dev)
        m.inner_model._pdrops = (0.0, 0.3, 0.0)
        return m

    agent = build_agent(cfg, model, device=dev, lr=1e-4)
    rng = np.random.default_rng(0)
    agent.get_scaler(Scaler(rng.standard_normal((64, cfg.obs_dim)).astype(np.float32),
                            rng.standard_normal((64, cfg.act_dim)).astype(np.float32), True, dev))
    agent.set_bounds(agent.scaler)
    torch.manual_seed(0)
    batch = {"observation": torch.randn(B, cfg.obs_seq_len, cfg.obs_dim, device=dev),
             "action": torch.tanh(torch.randn(B, cfg.obs_seq_len, cfg.act_dim, device=dev)),
             "goal_observation": torch.randn(B, cfg.goal_seq_len, cfg.obs_dim, device=dev)}
    for _ in range(20):
        agent.train_step(batch)
    torch.cuda.synchronize()
    N = 100
    t0 = time.perf_counter()
    for _ in range(N):
        agent.train_step(batch)
    torch.cuda.synchronize()
    print(f"train_step (with loss.item()): {(time.perf_counter() - t0) / N * 1e3:.3f} ms")

    # the same steps without the host read of the loss: the host may run ahead of the GPU
    import beso_amd.agents.diffusion_agents.beso_agent as BA
    item = torch.Tensor.item
    try:
        torch.Tensor.item = lambda self: 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            agent.train_step(batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    finally:
        torch.Tensor.item = item
    print(f"train_step without .item(): host enqueue {(t1 - t0) / N * 1e3:.3f} ms per step, wall {(t2 - t0) / N * 1e3:.3f} ms per step")

    # pieces of the host side
    state, action, goal = agent.process_batch(batch, predict=False)
    noise = torch.randn_like(action)
    sigma = agent.make_sample_density()(shape=(len(action),), device=dev)
    step = agent.model.hip_train_step(state, action, goal, noise, sigma)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        step.run(state, action, goal, noise, sigma, seed=1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"HipTrainStep.run: host enqueue {(t1 - t0) / N * 1e3:.3f} ms, wall {(t2 - t0) / N * 1e3:.3f} ms per call")
    t0 = time.perf_counter()
    for _ in range(N):
        noise = torch.randn_like(action)
        sigma = agent.make_sample_density()(shape=(len(action),), device=dev)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"noise + sigma draws: host {(t1 - t0) / N * 1e3:.3f} ms")
    t0 = time.perf_counter()
    for _ in range(N):
        agent.process_batch(batch, predict=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"process_batch: host {(t1 - t0) / N * 1e3:.3f} ms")
    t0 = time.perf_counter()
    for _ in range(N):
        agent.optimizer.step(ema=agent.ema_helper)
        agent.lr_scheduler.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"optimizer.step + lr_scheduler.step: host {(t1 - t0) / N * 1e3:.3f} ms, wall {(t2 - t0) / N * 1e3:.3f} ms")
    t0 = time.perf_counter()
    for _ in range(N):
        int(torch.randint(0, 2 ** 31 - 1, (1,), device="cpu").item())
    print(f"seed draw: host {(time.perf_counter() - t0) / N * 1e3:.3f} ms")


if __name__ == "__main__" and "--pieces" not in sys.argv:
    main()


def pieces():
    """host time of the parts of HipTrainStep.run (kitchen, bf16)"""
    import ctypes as C
    from beso_amd import _lib
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = "cuda:0"
    cfg = O.SHAPES["kitchen"]
    w = O.make_weights(cfg, seed=0, std=0.02)
    m = build_model(cfg, w, "bf16", dev)
    m.inner_model._pdrops = (0.0, 0.3, 0.0)
    m.train()
    torch.manual_seed(0)
    state = torch.randn(B, cfg.obs_seq_len, cfg.obs_dim, device=dev)
    action = torch.tanh(torch.randn(B, cfg.obs_seq_len, cfg.act_dim, device=dev))
    goal = torch.randn(B, cfg.goal_seq_len, cfg.obs_dim, device=dev)
    noise = torch.randn_like(action)
    sigma = torch.rand(B, device=dev) * 0.9 + 0.05
    step = m.hip_train_step(state, action, goal, noise, sigma)
    lib = step.lib
    orig = lib.beso_loss_grad_streams
    acc = [0.0, 0]

    class Timed:
        def __call__(self, *a):
            t0 = time.perf_counter()
            r = orig(*a)
            acc[0] += time.perf_counter() - t0
            acc[1] += 1
            return r
    for plan, name in ((0, "one-launch forward"), (_lib.TRAIN_PLAN_PER_OP, "per-op forward")):
        from beso_amd.runtime import set_plan
        set_plan(train=plan)
        for _ in range(10):
            step.run(state, action, goal, noise, sigma, seed=1)
        torch.cuda.synchronize()
        step.lib = type("L", (), {"__getattr__": lambda s, k: Timed() if k == "beso_loss_grad_streams" else getattr(lib, k)})()
        acc[0], acc[1] = 0.0, 0
        N = 100
        t0 = time.perf_counter()
        for _ in range(N):
            step.run(state, action, goal, noise, sigma, seed=1)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        step.lib = lib
        print(f"{name}: run() host {(t1 - t0) / N * 1e3:.3f} ms, of which the beso_loss_grad call {acc[0] / acc[1] * 1e3:.3f} ms; "
              f"wall {(t2 - t0) / N * 1e3:.3f} ms")
    set_plan(train=0)


if __name__ == "__main__" and "--pieces" in sys.argv:
    sys.argv.remove("--pieces")
    pieces()
