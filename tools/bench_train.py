#!/usr/bin/env python3
"""BASELINE config 3 on one GPU's share: kitchen train_step (score-matching loss, backward, AdamW, EMA) at
1024 samples per step, through BesoAgent.train_step: HIP forward + backward (beso_loss_grad) and the fused
Adam(W) + EMA launch; `--autograd` times the tests' torch-autograd comparator of the same step beside it.
    python tools/bench_train.py [batch] [kitchen|block_push] [--autograd]"""
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
from beso_amd import synthetic as O  # noqa: E402
from _agent import build_agent  # noqa: E402
from beso_amd.networks.scaler.scaler_class import Scaler  # noqa: E402


def main():
    if "--per-op-forward" in sys.argv:          # A/B: every layer of the training forward through the per-op kernels
        sys.argv.remove("--per-op-forward")
        from beso_amd import _lib
        from beso_amd.runtime import set_plan
        set_plan(train=_lib.TRAIN_PLAN_PER_OP)
    if "--tail-forward" in sys.argv:            # ... or through the tile kernel whatever the batch size
        sys.argv.remove("--tail-forward")
        from beso_amd import _lib
        from beso_amd.runtime import set_plan
        set_plan(train=_lib.TRAIN_PLAN_TILES)
    if "--autograd" in sys.argv:
        sys.argv.remove("--autograd")
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from autograd_reference import install_autograd_training
        install_autograd_training()
    # one process per GPU under torchrun (python -m torch.distributed.run --nproc-per-node N tools/bench_train.py B):
    # every rank trains on its own B samples per step, gradients all-reduced (C1); single process otherwise
    from beso_amd import distributed as bdist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        bdist.init_from_env("nccl")
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(dev)
    name = sys.argv[2] if len(sys.argv) > 2 else "kitchen"
    cfg = O.SHAPES[name]
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    w = O.make_weights(cfg, seed=0, std=0.02)
    # the shipped dropouts (configs/franka_kitchen_main_config.yaml:56-57, configs/block_push_main_config.yaml:57-58)
    attn_p, resid_p = {"kitchen": (0.3, 0.0), "block_push": (0.05, 0.05)}.get(name, (0.0, 0.0))

    def model():
        m = build_model(cfg, w, "bf16", dev)
        inner = m.inner_model
        inner._pdrops = (0.0, attn_p, resid_p)
        for blk in inner.blocks:
            blk.attn.attn_drop.p = attn_p
            blk.attn.resid_drop.p = resid_p
            for mod in blk.mlp:
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = resid_p
        return m

    agent = build_agent(cfg, model, device=dev)
    rng = np.random.default_rng(0)
    agent.get_scaler(Scaler(rng.standard_normal((256, cfg.obs_dim)).astype(np.float32),
                            rng.standard_normal((256, cfg.act_dim)).astype(np.float32), True, dev))
    agent.set_bounds(agent.scaler)
    if world > 1:
        bdist.broadcast_parameters(agent.model.get_params(), src=0)          # C2: replicas start identical
        agent.ema_helper.load_shadow_params(agent.model.get_params())
    torch.manual_seed(1234 + rank)
    batch = {"observation": torch.randn(B, cfg.obs_seq_len, cfg.obs_dim), "action": torch.randn(B, cfg.obs_seq_len, cfg.act_dim),
             "goal_observation": torch.randn(B, cfg.goal_seq_len, cfg.obs_dim)}
    batch = {k: v.to(dev) for k, v in batch.items()}
    for _ in range(3):
        loss = agent.train_step(batch)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        loss = agent.train_step(batch)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = (time.perf_counter() - t0) / n
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank != 0:
        return
    B = B * world                                                              # samples per step of the whole job
    flops = 3.0 * cfg.flops_per_sample() * B
    print(json.dumps({"config": ("3: kitchen train_step" if name == "kitchen" else name + " train_step"), "n_gpus": world, "batch": B, "attn_pdrop": attn_p, "resid_pdrop": resid_p, "seconds_per_step": dt,
                      "samples_per_s": B / dt, "tflops_fwd_bwd": flops / dt / 1e12, "loss": loss,
                      "path": ("HIP forward/backward (bf16 operands)" if getattr(agent, "_hip_step", None) is not None
                               else "torch autograd fp32 forward/backward") + " + " + type(agent.optimizer).__name__ + " (+EMA)"}))


if __name__ == "__main__":
    main()
