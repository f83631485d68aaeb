"""N > 1 path on CPU: two processes, gloo backend.  Covers the gradient all-reduce of the training
step (C1), the weight broadcast (C2), batch sharding of the inference path, and that a 2-rank
train_step equals a single-process step on the concatenated batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_agent(seed):
    from oracle import beso_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_logic import build_agent, make_module
    from beso_amd.networks.scaler.scaler_class import Scaler
    cfg = O.TINY
    torch.manual_seed(seed)
    agent = build_agent(cfg, lambda: make_module(cfg))
    rng = np.random.default_rng(0)
    agent.get_scaler(Scaler(rng.standard_normal((40, cfg.obs_dim)).astype(np.float32),
                            rng.uniform(-1, 1, (40, cfg.act_dim)).astype(np.float32), False, "cpu"))
    return cfg, agent


def _batch(cfg, n, seed):
    g = torch.Generator().manual_seed(seed)
    return {"observation": torch.randn(n, cfg.obs_seq_len, cfg.obs_dim, generator=g),
            "goal_observation": torch.randn(n, cfg.goal_seq_len, cfg.obs_dim, generator=g),
            "action": torch.rand(n, cfg.obs_seq_len, cfg.act_dim, generator=g) * 2 - 1}


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from autograd_reference import install_autograd_training
    install_autograd_training()          # CPU host-logic test: the loss comes from the tests' autograd comparator
    from beso_amd import distributed as bdist
    assert bdist.init_from_env("gloo")
    assert bdist.world_size() == world and bdist.rank() == rank and bdist.is_distributed()
    # --- C1: flat gradient bucket = mean over ranks
    p = [torch.nn.Parameter(torch.full((3, 2), float(rank + 1))), torch.nn.Parameter(torch.zeros(5))]
    p[0].grad = torch.full((3, 2), float(rank + 1))
    p[1].grad = None                                   # a parameter that received no gradient on this rank
    bucket = bdist.GradientBucket(p)
    bucket.sync()
    assert torch.allclose(p[0].grad, torch.full((3, 2), 1.5)) and torch.allclose(p[1].grad, torch.zeros(5))
    p[0].grad = torch.full((3, 2), float(10 * (rank + 1)))
    bucket.sync(async_op=True)
    bucket.wait()
    assert torch.allclose(p[0].grad, torch.full((3, 2), 15.0))
    # --- C1 on the flat buffer the HIP training step writes (already scaled by 1/world by the kernels)
    flat_g = torch.full((7,), float(rank + 1) / world)
    bdist.all_reduce_sum(flat_g)
    assert torch.allclose(flat_g, torch.full((7,), 1.5))
    # --- the same exchange split into the early range and the rest (the overlapped form; no streams on the CPU)
    for rng in ((3, 6), (0, 4), (2, 7), (0, 7), (5, 5)):
        flat_g = torch.arange(7, dtype=torch.float32) * (rank + 1) / world
        bdist.all_reduce_sum_overlapped(flat_g, rng, None)
        assert torch.allclose(flat_g, torch.arange(7, dtype=torch.float32) * 1.5), rng
    # --- the two halves of the sharded exchange: reduce-scatter + all-gather == all-reduce, for n not a multiple of world
    for n in (7, 8, 1):
        g0 = torch.arange(n, dtype=torch.float32) * (rank + 1) / world
        s = bdist.shard_size(n, world)
        padded = torch.zeros(s * world)
        padded[:n] = g0
        shard = torch.empty(s)
        bdist.reduce_scatter_flat(padded, shard)
        lo, hi = bdist.shard_range_flat(n, world, rank)
        assert torch.allclose(shard[: hi - lo], (torch.arange(n, dtype=torch.float32) * 1.5)[lo:hi]), (n, shard)
        full = torch.empty(s * world)
        bdist.all_gather_flat(full, shard)
        assert torch.allclose(full[:n], torch.arange(n, dtype=torch.float32) * 1.5), n
    # ShardedExchange on a ragged parameter list: gradients reduced into the owned range, parameters / a flat state gathered
    torch.manual_seed(5)
    ps = [torch.nn.Parameter(torch.randn(3, 2)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 1))]
    ex = bdist.ShardedExchange(ps)
    assert ex.n == 13 and ex.S == 7 and (ex.lo, ex.hi) == ((0, 7) if rank == 0 else (7, 13))
    gpad = torch.zeros(ex.padded)
    gpad[:13] = torch.arange(13, dtype=torch.float32) * (rank + 1) / world
    ex.reduce_scatter_grads(gpad)
    assert torch.allclose(gpad[ex.lo:ex.hi], (torch.arange(13, dtype=torch.float32) * 1.5)[ex.lo:ex.hi])
    flat_p = torch.cat([q.detach().reshape(-1) for q in ps])                  # same seed: replicas agree to start with
    with torch.no_grad():                                                     # every rank "updates" only what it owns
        for i, a, b, at in ex._pieces(ex.lo, ex.hi):
            ps[i].reshape(-1)[a:b] += 100.0 * (rank + 1)
    ex.all_gather_params()
    want = flat_p.clone()
    want[:7] += 100.0
    want[7:] += 200.0
    assert torch.allclose(torch.cat([q.detach().reshape(-1) for q in ps]), want)
    state = torch.full((13,), -1.0)
    state[ex.lo:ex.hi] = float(rank + 1)
    ex.all_gather_flat_state(state)
    assert torch.equal(state, torch.cat([torch.full((7,), 1.0), torch.full((6,), 2.0)]))
    # --- C2: broadcast makes replicas identical
    cfg, agent = _make_agent(seed=100 + rank)           # different init per rank on purpose
    bdist.broadcast_parameters(agent.model.get_params(), src=0)
    agent.ema_helper.load_shadow_params(agent.model.get_params())
    flat = torch.cat([q.detach().reshape(-1) for q in agent.model.get_params()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert torch.equal(gathered[0], gathered[1])
    # --- _sync_replicas (train_agent's first act): same weights everywhere, EMA re-seeded, per-rank random streams
    cfg2, agent2 = _make_agent(seed=200 + rank)
    torch.manual_seed(4242)                              # the scripts seed every rank alike
    agent2._sync_replicas()
    flat2 = torch.cat([q.detach().reshape(-1) for q in agent2.model.get_params()])
    g2 = [torch.zeros_like(flat2) for _ in range(world)]
    dist.all_gather(g2, flat2)
    assert torch.equal(g2[0], g2[1])
    assert torch.equal(torch.cat([q.reshape(-1) for q in agent2.ema_helper.shadow_params]), flat2)
    draw = torch.randn(4)
    draws = [torch.zeros(4) for _ in range(world)]
    dist.all_gather(draws, draw)
    assert not torch.equal(draws[0], draws[1]), "ranks would draw identical noise / sigma / masks"
    # --- one data-parallel training step: each rank sees its shard of the global batch
    full = _batch(cfg, 8, seed=7)
    lo, hi = bdist.shard_range(8, world, rank)
    shard = {k: v[lo:hi] for k, v in full.items()}
    torch.manual_seed(1234)                             # same noise / sigma stream on both ranks ...
    noise = torch.randn(8, cfg.obs_seq_len, cfg.act_dim)
    sigma = torch.rand(8) * 0.9 + 0.05
    real_randn_like, real_density = torch.randn_like, agent.make_sample_density
    torch.randn_like = lambda t: noise[lo:hi].clone()   # ... sliced like the batch
    agent.make_sample_density = lambda: (lambda shape, device: sigma[lo:hi].clone())
    os.environ["BESO_AMD_C1"] = os.environ.get("TEST_C1_MODE", "overlap")
    try:
        loss = agent.train_step(shard)
    finally:
        torch.randn_like = real_randn_like
        agent.make_sample_density = real_density
    after = torch.cat([q.detach().reshape(-1) for q in agent.model.get_params()])
    gathered = [torch.zeros_like(after) for _ in range(world)]
    dist.all_gather(gathered, after)
    assert torch.equal(gathered[0], gathered[1]), "replicas diverged after the all-reduced step"
    if rank == 0:
        torch.save({"params": after, "loss": loss, "noise": noise, "sigma": sigma}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("c1_mode", ["overlap", "sharded", "flat"])
def test_two_rank_gloo_training_step_matches_single_process(tmp_path, autograd_training, c1_mode, monkeypatch):
    out = str(tmp_path / "dp.pt")
    port = _free_port()
    monkeypatch.setenv("TEST_C1_MODE", c1_mode)          # (inherited by the spawned ranks)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    dp = torch.load(out)
    # single process, whole batch, same init (rank 0's) and the same noise / sigma
    sys.path.insert(0, ROOT)
    cfg, agent = _make_agent(seed=100)
    full = _batch(cfg, 8, seed=7)
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t: dp["noise"].clone()
    agent.make_sample_density = lambda: (lambda shape, device: dp["sigma"].clone())
    try:
        single_loss = agent.train_step(full)
    finally:
        torch.randn_like = real_randn_like
    assert abs(single_loss - dp["loss"]) < 1e-5 * abs(single_loss)          # the data-parallel step reports the global mean
    single = torch.cat([q.detach().reshape(-1) for q in agent.model.get_params()])
    # mean of per-shard gradients == gradient of the global mean loss (equal shard sizes); AdamW sees the same grads
    # (the first AdamW step is lr * g / (|g| + eps): fp32 summation-order noise in g shows up at the 1e-7 level)
    assert torch.allclose(single, dp["params"], rtol=2e-5, atol=2e-6)


def test_shard_range_covers_the_batch():
    from beso_amd.distributed import shard_range
    for total in (1, 7, 8, 4096, 4097):
        for n in (1, 2, 3, 8):
            spans = [shard_range(total, n, i) for i in range(n)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _loop_worker(rank, world, port, out, mode, c1, use_ema):
    """Two ranks run BesoAgent's own training loops with test metrics that DIFFER between the ranks (different evaluation
    noise / test shards in a real job; scripted here) and a rank whose test loader is empty."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), BESO_AMD_C1=c1)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from autograd_reference import install_autograd_training
    install_autograd_training()
    from beso_amd import distributed as bdist
    assert bdist.init_from_env("gloo")
    cfg, agent = _make_agent(seed=300 + rank)
    agent.use_ema = use_ema
    agent.working_dir = os.path.dirname(out)
    script = {0: [1.0, 2.0, 0.5, 9.0, 9.0], 1: [3.0, 0.2, 4.0, 9.0, 9.0]}[rank] if mode == "steps" else \
             {0: [1.0, 0.9, 0.8, 0.7, 0.6], 1: [1.0, 1.5, 2.0, 2.5, 3.0]}[rank]
    calls = {"eval": 0, "store": 0}

    def fake_evaluate(batch):
        agent._complete_ema()                                  # (what the real evaluate reaches through _ema_scope)
        v = script[min(calls["eval"], len(script) - 1)]
        calls["eval"] += 1
        return v

    real_store = agent.store_model_weights

    def counting_store(path):
        calls["store"] += 1
        return real_store(path)

    agent.evaluate = fake_evaluate
    agent.store_model_weights = counting_store
    if os.environ.get("TEST_FORCE_EMA_PARTIAL") == "1":
        # On CPU the step never leaves the EMA shadow partial (that needs the HIP step + FusedAdam), so _complete_ema would be
        # a no-op and the collective ORDER of the loops would go untested: mark the shadow partial after every step, which
        # makes every _complete_ema a real all-gather over gloo (a rank without test data must still reach it).
        real_step = agent.train_step

        def step_leaving_the_shadow_partial(batch):
            v = real_step(batch)
            agent._ema_partial = True
            return v
        agent.train_step = step_leaving_the_shadow_partial
        gathers = {"n": 0}
        real_gather = agent._sharded().all_gather_flat_state

        def counting_gather(flat):
            gathers["n"] += 1
            return real_gather(flat)
        agent._sharded().all_gather_flat_state = counting_gather
    train = [_batch(cfg, 4, seed=11 + rank), _batch(cfg, 4, seed=21 + rank)]
    test = [_batch(cfg, 4, seed=31)] if (rank == 0 or mode == "epochs") else []      # steps mode: rank 1 has no test data
    if mode == "steps":
        agent.train_method, agent.max_train_steps, agent.eval_every_n_steps = "steps", 6, 2
    else:
        agent.train_method, agent.epochs, agent.patience, agent.eval_every_n_steps = "epochs", 5, 0, 100
        agent.epochs_no_improvement = 0
    agent.train_agent(train, test)
    after = torch.cat([q.detach().reshape(-1) for q in agent.model.get_params()])
    gathered = [torch.zeros_like(after) for _ in range(world)]
    dist.all_gather(gathered, after)
    assert torch.equal(gathered[0], gathered[1]), "replicas diverged"
    if os.environ.get("TEST_FORCE_EMA_PARTIAL") == "1":
        n = torch.tensor([float(gathers["n"])], dtype=torch.float64)
        ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n)
        assert ns[0].item() == ns[1].item() and ns[0].item() >= 1, ns     # the same all-gathers on both ranks, and they happened
    counts = torch.tensor([calls["eval"], calls["store"], agent.steps], dtype=torch.float64)
    both = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(both, counts)
    if rank == 0:
        torch.save({"counts": [b.tolist() for b in both]}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["steps", "epochs"])
def test_a_rank_without_test_data_still_joins_the_ema_all_gather(tmp_path, mode, monkeypatch):
    """Sharded C1 leaves the EMA shadow partial; the all-gather that completes it must be reached by EVERY rank at the same
    point of the loops -- also by a rank whose test loader is empty and therefore never calls evaluate() (steps mode below).
    The loops issue it themselves in front of the evaluation list; with the gather only inside evaluate() this test hangs
    (rank 1 goes straight to job_mean's all-reduce)."""
    monkeypatch.setenv("TEST_FORCE_EMA_PARTIAL", "1")
    out = str(tmp_path / "loops.pt")
    mp.spawn(_loop_worker, args=(2, _free_port(), out, mode, "sharded", True), nprocs=2, join=True)
    c = torch.load(out)["counts"]
    assert c[0][1:] == c[1][1:], c


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode,c1,use_ema", [("steps", "sharded", False), ("steps", "sharded", True), ("steps", "overlap", True),
                                             ("epochs", "sharded", True), ("epochs", "flat", False)])
def test_two_rank_training_loops_take_joint_decisions(tmp_path, mode, c1, use_ema):
    """BesoAgent.train_agent on two gloo ranks whose rank-local test MSEs point different ways (and, in steps mode, one rank
    without test data): checkpoints and early stopping follow the JOB-WIDE mean, so both ranks store the same number of
    times, stop at the same epoch and end with identical weights -- with the sharded C1 exchange (whose checkpoint path
    holds an all-gather of the EMA shadow) and without EMA evaluation included.  A rank deciding alone would hang here."""
    out = str(tmp_path / "loops.pt")
    mp.spawn(_loop_worker, args=(2, _free_port(), out, mode, c1, use_ema), nprocs=2, join=True)
    c = torch.load(out)["counts"]
    assert c[0][1:] == c[1][1:], c                      # same number of checkpoints, same number of optimizer steps
    if mode == "steps":
        # job means of the scripted evaluations: 2.0, 1.1, 2.25 -> two improvements + the final store; 6 steps
        assert c[0][1] == 3 and c[0][2] == 6, c
    else:
        # job means 1.0, 1.2, ...: epoch 0 improves, epoch 1 does not -> both stop there (rank 0 alone never would)
        assert c[0][0] == 2 and c[1][0] == 2 and c[0][1] == 2, c


def _bench_train_worker(rank, world, port, c1):
    """bench.py's `--workload train` leg (run_train: BesoAgent.train_step on every rank's own batch, C1 as configured) on two
    gloo ranks: after 20 + warm-up steps the replicas hold identical parameters and an identical EMA shadow."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      BESO_AMD_C1=c1)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from autograd_reference import install_autograd_training
    install_autograd_training()
    import argparse
    import bench
    from beso_amd import distributed as bdist
    assert bdist.init_from_env("gloo")
    torch.set_num_threads(4)
    args = argparse.Namespace(config="block_push", batch=2, steps=20, warmup=2, settle_ms=0.0, precision="fp32", c1_overlap=1,
                              keep_agent=[])
    res = bench.run_train(args, world, rank, "cpu")
    assert (res is not None) == (rank == 0)
    if rank == 0:
        assert res["n_gpus"] == 2 and res["steps"] == 20 and np.isfinite(res["loss"])
    agent = args.keep_agent[0]
    agent._complete_ema()                                    # (sharded C1: the shadow is gathered before it is read)
    for name, flat in (("parameters", torch.cat([q.detach().reshape(-1) for q in agent.model.get_params()])),
                       ("EMA shadow", agent.ema_helper._flat.detach().reshape(-1).clone())):
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1]), f"{name} differ between the ranks after the run ({c1})"
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("c1", ["overlap", "sharded", "flat"])
def test_bench_train_workload_keeps_two_gloo_replicas_identical(c1):
    mp.spawn(_bench_train_worker, args=(2, _free_port(), c1), nprocs=2, join=True)
