"""Parity tests proper: the HIP path (through the C ABI, libbeso_hip.so) against the CPU oracle and
the reference-generated golden vectors, on a real MI355X.

Tolerances (relative to max |reference| over the output tensor):
  fp32 mode  (exact-fp32 MFMA)      : 2e-5  -- the north-star 1e-4 bound with margin
  bf16 mode  (bf16 MFMA, fp32 acc)  : 3e-2 on a single denoiser call with synthetic weights; the
              measured values are printed and recorded in DESIGN.md (typically 2e-3 .. 8e-3).
"""
import functools
import os

import numpy as np
import pytest
import torch

from oracle import beso_oracle as O
from conftest import load_golden, weights_from_fixture, rel_err

pytestmark = pytest.mark.gpu

TOL = {"fp32": 2e-5, "bf16": 3e-2, "bf16x3": 1e-4, "fp16": 4e-3}      # bf16x3: the north-star bound itself (measured 7e-6 .. 3e-5)
# bf16 against the reference's vectors, per fixture: twice what the fused kernel measures on MI355X (round 2:
# 7.3e-3, 1.02e-2, 9.3e-3; the per-op kernels of the non-fused shapes sit at 3e-3 .. 8e-3)
TOL_BF16_FIXTURE = {"kitchen_forward_std002.npz": 1.5e-2, "kitchen_forward_std008.npz": 2.1e-2, "block_push_forward.npz": 1.9e-2}
# fp16 operands (BESO_PREC_FP16) on the same vectors: twice what the kernel measures on MI355X (round 3)
TOL_FP16_FIXTURE = {"kitchen_forward_std002.npz": 4e-3, "kitchen_forward_std008.npz": 4e-3, "block_push_forward.npz": 4e-3,
                    "long_horizon_forward.npz": 4e-3}
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _fused_kernels_unless_a_test_asks_for_the_small_batch_path():
    """The fixtures of this file are small batches (4 ... 67 samples), which the library by itself now serves on the chip-wide
    small-batch path (round 5, small.hip).  The tests written for the one-launch / block / per-op kernels keep asking for those
    (BESO_PLAN_FUSED: 'as if the small-batch path did not exist'; set_level / set_instances add their own bits); the tests of
    the small-batch path -- the library's default included -- clear the hint themselves."""
    from beso_amd import _lib
    from beso_amd.runtime import set_plan
    set_plan(forward=_lib.PLAN_FUSED, train=0)
    yield
    set_plan(forward=0, train=0)


def count_site_launches(site, fn):
    """Runs fn() with the launch-site timer on `site`; returns the number of timed regions it recorded."""
    from beso_amd import _lib
    import ctypes as C
    lib = _lib.load()
    lib.beso_profile_enable(_lib.SITES[site])
    try:
        fn()
        torch.cuda.synchronize()
        ms, n = C.c_double(0.0), C.c_int(0)
        assert lib.beso_profile_read(C.byref(ms), C.byref(n)) == 0
    finally:
        lib.beso_profile_enable(0)
    return n.value


def set_level(lvl):
    """Which kernels the forward calls of this thread may use (BESO_PLAN_* hints carried by every call): 2 the one-launch
    kernel where the shape has it (the library's own choice), 1 at most the block kernels, 0 the per-op kernels."""
    from beso_amd import _lib
    from beso_amd.runtime import forward_hints, set_plan
    set_plan(forward=(forward_hints() & ~0x30) | {2: 0, 1: _lib.PLAN_BLOCKS, 0: _lib.PLAN_PER_OP}[lvl])


def set_instances(limit):
    """Which instance of the one-launch kernel: 0 -> eight samples per workgroup at every batch size, 1 << 20 -> two,
    512 -> the library's own choice (two up to 512 samples, four up to 1024, eight beyond)."""
    from beso_amd import _lib
    from beso_amd.runtime import forward_hints, set_plan
    spw = _lib.PLAN_SPW8 if limit == 0 else (_lib.PLAN_SPW2 if limit >= (1 << 20) else 0)
    set_plan(forward=(forward_hints() & ~0x300) | spw)


def set_train_tail(on):
    """Forward half of the bf16 training step: 0 per-op kernels, 1 the library's choice, 2 the tile kernel always."""
    from beso_amd import _lib
    from beso_amd.runtime import set_plan
    set_plan(train={0: _lib.TRAIN_PLAN_PER_OP, 1: 0, 2: _lib.TRAIN_PLAN_TILES}[on])


def count_fused_launches(fn):
    """Runs fn() with the launch-site timer on the fused kernel's site; returns the number of launches it recorded."""
    from beso_amd import _lib
    import ctypes as C
    lib = _lib.load()
    lib.beso_profile_enable(_lib.SITES["fused_layer"])
    try:
        fn()
        torch.cuda.synchronize()
        ms, n = C.c_double(0.0), C.c_int(0)
        assert lib.beso_profile_read(C.byref(ms), C.byref(n)) == 0
    finally:
        lib.beso_profile_enable(0)
    return n.value


def make_module(cfg, w=None, precision="fp32", **kw):
    from beso_amd.agents.diffusion_agents.k_diffusion.score_gpts import DiffusionGPT
    from beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser
    inner = functools.partial(
        DiffusionGPT, state_dim=cfg.obs_dim, device=DEV, goal_conditioned=cfg.goal_conditioned,
        action_dim=cfg.act_dim, embed_dim=cfg.embed_dim, embed_pdrob=0.0, attn_pdrop=kw.get("attn_pdrop", 0.0),
        resid_pdrop=0.0, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=cfg.goal_seq_len,
        obs_seq_len=cfg.obs_seq_len, sigma_vocab_size=3, time_embedding_fn=None, goal_drop=0.0,
        linear_output=cfg.linear_output, precision=precision)
    m = GCDenoiser(inner, sigma_data=cfg.sigma_data)
    if w is not None:
        sd = m.state_dict()
        for k, v in w.items():
            sd[k] = torch.from_numpy(v.copy())
        m.load_state_dict(sd)
    return m.to(DEV).eval()


def G(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def _weights(fx, cfg):
    w = weights_from_fixture(fx)
    return w if w else O.make_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"]))


def test_native_library_is_what_runs():
    """The forward must come from libbeso_hip.so: it is loaded in this process and torch ops are
    not a fallback (a CPU tensor raises)."""
    from beso_amd import _lib
    lib = _lib.load()
    assert b"gfx950" in lib.beso_version()
    from beso_amd import build as B
    assert B.is_current(), "libbeso_hip.so was not built from the sources in this tree (stale binary)"
    maps = open("/proc/self/maps").read()
    assert "libbeso_hip.so" in maps
    m = make_module(O.TINY, O.make_weights(O.TINY))
    with torch.no_grad(), pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 7), torch.zeros(1, 3, 3), torch.zeros(1, 2, 7), torch.ones(1))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("fixture,cfg_name", [
    ("tiny_forward.npz", "tiny"), ("tiny_mlp_head_forward.npz", "tiny_mlp_head"),
    ("tiny_nogoal_forward.npz", "tiny_nogoal"), ("kitchen_forward_std002.npz", "kitchen"),
    ("kitchen_forward_std008.npz", "kitchen"), ("block_push_forward.npz", "block_push"),
    ("long_horizon_forward.npz", "long_horizon")])
def test_forward_vs_reference_vectors(fixture, cfg_name, precision):
    """GCDenoiser.forward / DiffusionGPT.forward, t in {1, W/2, W}, cond and uncond."""
    fx = load_golden(fixture)
    cfg = O.CONFIGS[cfg_name]
    m = make_module(cfg, _weights(fx, cfg), precision)
    worst = 0.0
    with torch.no_grad():
        for t in fx["ts"]:
            p = f"t{int(t)}::"
            s, a, g, sg = (G(fx[p + k]) for k in ("state", "action", "goal", "sigma"))
            e1 = rel_err(m(s, a, g, sg).cpu().numpy(), fx[p + "denoised"])
            e2 = rel_err(m(s, a, g, sg, uncond=True).cpu().numpy(), fx[p + "denoised_uncond"])
            e3 = rel_err(m.inner_model(s, a, g, sg).cpu().numpy(), fx[p + "inner"])
            worst = max(worst, e1, e2, e3)
    print(f"[parity] {fixture} {precision}: max rel err {worst:.3e}")
    assert worst < (TOL_BF16_FIXTURE.get(fixture, TOL[precision]) if precision == "bf16" else TOL[precision])


@pytest.mark.parametrize("fixture,cfg_name", [("kitchen_forward_std002.npz", "kitchen"), ("kitchen_forward_std008.npz", "kitchen"),
                                              ("block_push_forward.npz", "block_push"), ("long_horizon_forward.npz", "long_horizon")])
def test_fp16_forward_through_the_one_launch_kernel(fixture, cfg_name):
    """BESO_PREC_FP16: the one-launch kernel with fp16 GEMM operands (v_mfma_f32_16x16x32_f16: the bf16 rate, three more
    mantissa bits) against the reference's vectors -- every call one launch at the fused kernel's site; the error is
    printed beside what the bf16 instance measures on the same vectors (7e-3 .. 1e-2) and held to TOL_FP16_FIXTURE."""
    fx = load_golden(fixture)
    cfg = O.CONFIGS[cfg_name]
    worst = {}
    for precision in ("fp16", "bf16"):
        m = make_module(cfg, _weights(fx, cfg), precision)
        worst[precision] = 0.0
        calls = 0

        def run():
            nonlocal calls
            for t in fx["ts"]:
                p = f"t{int(t)}::"
                s, a, g, sg = (G(fx[p + k]) for k in ("state", "action", "goal", "sigma"))
                e1 = rel_err(m(s, a, g, sg).cpu().numpy(), fx[p + "denoised"])
                e2 = rel_err(m(s, a, g, sg, uncond=True).cpu().numpy(), fx[p + "denoised_uncond"])
                e3 = rel_err(m.inner_model(s, a, g, sg).cpu().numpy(), fx[p + "inner"])
                worst[precision] = max(worst[precision], e1, e2, e3)
                calls += 3

        with torch.no_grad():
            launches = count_fused_launches(run)
        assert launches == calls
    print(f"[parity] {fixture}: fp16 {worst['fp16']:.3e}  (bf16 {worst['bf16']:.3e})")
    assert worst["fp16"] < TOL_FP16_FIXTURE[fixture]


def test_fp16_sampler_loops_and_rejections():
    """Sampler loops in the fp16 mode (one launch per call) against the reference's sampler outputs, classifier-free
    guidance included; shapes without the one-launch kernel are refused (the mode has no per-op form)."""
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    fns = {"ddim": ks.sample_ddim, "euler": ks.sample_euler, "heun": ks.sample_heun}
    for fixture, cfg_name in [("kitchen_samplers.npz", "kitchen"), ("block_push_heun_cfg.npz", "block_push"),
                              ("long_horizon_euler100.npz", "long_horizon")]:
        fx = load_golden(fixture)
        cfg = O.CONFIGS[cfg_name]
        m = make_module(cfg, _weights(fx, cfg), "fp16")
        lam = float(fx["cond_lambda"])
        model = m if lam < 0 else ClassifierFreeSampleModel(m, lam)
        for key in sorted(k[:-5] for k in fx if k.endswith("::out")):
            n = int(key.split("_")[-2])
            sampler = key[: key.index(f"_{n}_")]
            if sampler not in fns:
                continue
            out = {}
            with torch.no_grad():
                launches = count_fused_launches(lambda: out.__setitem__(0, fns[sampler](
                    model, G(fx["state"]), G(fx["x_t"]), G(fx["goal"]), torch.from_numpy(fx[key + "::sigmas"]), disable=True)))
            err = rel_err(out[0].cpu().numpy(), fx[key + "::out"])
            print(f"[parity] {fixture}:{key} fp16: {err:.3e} ({launches} launch)")
            assert launches == 1 and err < TOL["fp16"], key
    cfg = O.TINY
    m = make_module(cfg, O.make_weights(cfg), "fp16")
    s, g, a = (G(v) for v in O.make_inputs(cfg, 2, seed=0))
    with torch.no_grad(), pytest.raises(ValueError):
        m(s, a, g, G(np.full(2, 0.5, np.float32)))


@pytest.mark.parametrize("fixture,cfg_name", [("kitchen_forward_std002.npz", "kitchen"), ("kitchen_forward_std008.npz", "kitchen"),
                                              ("block_push_forward.npz", "block_push")])
def test_bf16x3_forward_through_the_fused_kernel(fixture, cfg_name):
    """The north-star tolerance (1e-4 of the reference) met by the BENCHMARKED kernel: the split-bf16 instance of
    layers_kernel (BESO_PREC_BF16X3) against the reference's own outputs (score_gpts.py:272-358, score_wrappers.py:81-96),
    t in {1, W/2, W}, conditional and unconditional, with and without preconditioning -- and every call must be exactly
    one launch at the fused kernel's launch site (the mode has no per-op form to fall back to)."""
    fx = load_golden(fixture)
    cfg = O.CONFIGS[cfg_name]
    m = make_module(cfg, _weights(fx, cfg), "bf16x3")
    worst, calls = 0.0, 0

    def run():
        nonlocal worst, calls
        for t in fx["ts"]:
            p = f"t{int(t)}::"
            s, a, g, sg = (G(fx[p + k]) for k in ("state", "action", "goal", "sigma"))
            e1 = rel_err(m(s, a, g, sg).cpu().numpy(), fx[p + "denoised"])
            e2 = rel_err(m(s, a, g, sg, uncond=True).cpu().numpy(), fx[p + "denoised_uncond"])
            e3 = rel_err(m.inner_model(s, a, g, sg).cpu().numpy(), fx[p + "inner"])
            worst = max(worst, e1, e2, e3)
            calls += 3

    with torch.no_grad():
        launches = count_fused_launches(run)
    print(f"[parity] {fixture} bf16x3 (layers_kernel, {launches} launches for {calls} calls): max rel err {worst:.3e}")
    assert launches == calls
    assert worst < TOL["bf16x3"]


def test_bf16x3_long_horizon_through_the_split_bf16_block_kernels():
    """BASELINE config 5's shape (D = 512, 67 tokens) in the 1e-4 mode: a sample's tokens do not fit a split-bf16 workgroup's
    LDS twice over, so BF16X3 runs the two-launch-per-layer form -- lin_block_x3_kernel (out-projection + residual -> LN2 ->
    MLP -> the next layer's LN1 + q/k/v in split-bf16 arithmetic, residual in registers) around the exact-fp32 attention
    kernel -- against the reference's forward vectors and its 100-step Euler run: <= 1e-4, with the launch sites asserted
    (L + 1 block launches per forward), classifier-free pairs and short windows included."""
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    cfg = O.CONFIGS["long_horizon"]
    fx = load_golden("long_horizon_forward.npz")
    m = make_module(cfg, _weights(fx, cfg), "bf16x3")
    worst, calls = 0.0, 0

    def run():
        nonlocal worst, calls
        for t in fx["ts"]:
            p = f"t{int(t)}::"
            s, a, g, sg = (G(fx[p + k]) for k in ("state", "action", "goal", "sigma"))
            e1 = rel_err(m(s, a, g, sg).cpu().numpy(), fx[p + "denoised"])
            e2 = rel_err(m(s, a, g, sg, uncond=True).cpu().numpy(), fx[p + "denoised_uncond"])
            e3 = rel_err(m.inner_model(s, a, g, sg).cpu().numpy(), fx[p + "inner"])
            worst = max(worst, e1, e2, e3)
            calls += 3

    with torch.no_grad():
        launches = count_fused_launches(run)
    print(f"[parity] long_horizon_forward.npz bf16x3 (block kernels, {launches} launches for {calls} calls): {worst:.3e}")
    assert launches == calls * (cfg.n_layers + 1) and worst < TOL["bf16x3"]
    for fixture in ("long_horizon_euler.npz", "long_horizon_euler100.npz"):
        fe = load_golden(fixture)
        me = make_module(cfg, _weights(fe, cfg), "bf16x3")
        for key in sorted(k[:-5] for k in fe if k.endswith("::out")):
            with torch.no_grad():
                out = ks.sample_euler(me, G(fe["state"]), G(fe["x_t"]), G(fe["goal"]), torch.from_numpy(fe[key + "::sigmas"]), disable=True)
            err = rel_err(out.cpu().numpy(), fe[key + "::out"])
            print(f"[parity] {fixture}:{key} bf16x3: {err:.3e}")
            assert err < TOL["bf16x3"], key
    # classifier-free pairs and a ragged batch against the oracle
    w = O.make_weights(cfg, seed=21, std=0.03)
    mo = make_module(cfg, w, "bf16x3")
    with torch.no_grad():
        for B, t in [(1, cfg.obs_seq_len), (5, 3), (37, 16)]:
            s_np, g_np, a_np = O.make_inputs(cfg, B, seed=100 * B + t, t=t)
            sg_np = np.linspace(0.06, 1.0, B).astype(np.float32)
            out = ClassifierFreeSampleModel(mo, 1.5)(G(s_np), G(a_np), G(g_np), G(sg_np))
            e = rel_err(out.cpu().numpy(), O.denoise_cfg(w, cfg, s_np, a_np, g_np, sg_np, 1.5))
            assert e < TOL["bf16x3"], (B, t, e)


def test_bf16x3_sampler_loops_and_cfg_through_the_fused_kernel():
    """Sampler loops (DDIM 3 / 10, Euler 10, Heun 5, DPM variants; block-push Heun-50 x classifier-free guidance: 198
    score-net forwards per sample) in the split-bf16 mode against the reference's sampler outputs: <= 1e-4."""
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    fns = {"ddim": ks.sample_ddim, "euler": ks.sample_euler, "heun": ks.sample_heun, "dpmpp_2m": ks.sample_dpmpp_2m,
           "dpm": ks.sample_dpm_2, "dpmpp_2s": ks.sample_dpmpp_2s}
    for fixture, cfg_name in [("kitchen_samplers.npz", "kitchen"), ("block_push_heun_cfg.npz", "block_push"),
                              ("block_push_cfg.npz", "block_push")]:
        fx = load_golden(fixture)
        cfg = O.CONFIGS[cfg_name]
        m = make_module(cfg, _weights(fx, cfg), "bf16x3")
        if fixture == "block_push_cfg.npz":
            with torch.no_grad():
                for lam in fx["lambdas"]:
                    out = ClassifierFreeSampleModel(m, float(lam))(G(fx["state"]), G(fx["action"]), G(fx["goal"]), G(fx["sigma"]))
                    err = rel_err(out.cpu().numpy(), fx[f"lam{float(lam)}"])
                    print(f"[parity] {fixture}:lambda={float(lam):g} bf16x3: {err:.3e}")
                    assert err < TOL["bf16x3"]
            continue
        lam = float(fx["cond_lambda"])
        model = m if lam < 0 else ClassifierFreeSampleModel(m, lam)
        for key in sorted(k[:-5] for k in fx if k.endswith("::out")):
            n = int(key.split("_")[-2])
            sampler = key[: key.index(f"_{n}_")]
            with torch.no_grad():
                out = fns[sampler](model, G(fx["state"]), G(fx["x_t"]), G(fx["goal"]), torch.from_numpy(fx[key + "::sigmas"]),
                                   disable=True)
            err = rel_err(out.cpu().numpy(), fx[key + "::out"])
            print(f"[parity] {fixture}:{key} bf16x3: {err:.3e}")
            assert err < TOL["bf16x3"], key


def test_bf16x3_rejects_shapes_without_a_fused_instance():
    """BF16X3 is an instance of the fused kernel only: shapes that kernel does not serve raise instead of silently
    running something else."""
    cfg = O.TINY
    m = make_module(cfg, O.make_weights(cfg), "bf16x3")
    s, g, a = (G(v) for v in O.make_inputs(cfg, 2, seed=0))
    with torch.no_grad(), pytest.raises(ValueError):
        m(s, a, g, G(np.full(2, 0.5, np.float32)))


@pytest.mark.parametrize("cfg_name", ["kitchen", "block_push"])
def test_fused_bf16_kernel_against_the_per_op_bf16_kernels(cfg_name):
    """The one-launch bf16 kernel against the per-op bf16 path of the same library (LayerNorm, GEMM, attention kernels:
    same arithmetic type, different rounding points -- the fused kernel folds LayerNorm's gamma into the weights before
    the bf16 rounding and keeps the residual in fp32 accumulators).  Both carry independent bf16 rounding errors of the
    same size: measured on MI355X (std 0.02) fused-vs-oracle 8.9e-4 / 5.9e-4, per-op-vs-oracle 8.6e-4 / 6.4e-4,
    fused-vs-per-op 1.1e-3 / 8.4e-4 (kitchen / block-push).  Held to: the fused kernel is no less accurate than the
    per-op path (x1.5), and the two differ by no more than twice the larger of their own errors."""
    from beso_amd import _lib
    lib = _lib.load()
    cfg = O.CONFIGS[cfg_name]
    for std, bound in ((0.02, 2e-3), (0.08, 2e-2)):
        wts = O.make_weights(cfg, seed=3, std=std)
        m = make_module(cfg, wts, "bf16")
        s_np, g_np, a_np = O.make_inputs(cfg, 256, seed=9)
        sg_np = np.linspace(0.05, 1.0, 256).astype(np.float32)
        s, a, g, sg = G(s_np), G(a_np), G(g_np), G(sg_np)
        ref = O.denoise(wts, cfg, s_np, a_np, g_np, sg_np)
        outs = {}
        try:
            with torch.no_grad():
                for lvl in (2, 0):
                    set_level(lvl)
                    n = count_fused_launches(lambda: outs.__setitem__(lvl, m(s, a, g, sg).cpu().numpy()))
                    assert n == (1 if lvl == 2 else 0)
        finally:
            set_level(2)
        e_f, e_p, e_fp = rel_err(outs[2], ref), rel_err(outs[0], ref), rel_err(outs[2], outs[0])
        print(f"[parity] {cfg_name} std={std}: fused-vs-oracle {e_f:.3e} per-op-vs-oracle {e_p:.3e} fused-vs-per-op {e_fp:.3e}")
        assert e_f < bound and e_f < 1.5 * e_p + 1e-4
        assert e_fp < 2.0 * max(e_f, e_p)


@pytest.mark.parametrize("B,t", [(1, 32), (3, 32), (37, 16), (5, 1), (2, 31)])
def test_long_sequence_layers_kernel_against_block_kernels_and_oracle(B, t):
    """Long-horizon shape (D = 512, up to 67 tokens: BASELINE config 5), bf16: the whole network as ONE launch -- a sample
    per workgroup in five token tiles, attention core with one query tile per wave (layers_kernel, CORE = 1) -- against the
    two-launches-per-layer form of the same library (attention kernel + tail block) and against the oracle, for full, short
    and odd windows (T = 67, 35, 5, 65 tokens) and batch sizes that leave workgroups empty-handed nowhere (one sample
    each).  Launch sites asserted: 1 fused launch at level 2, one tail block per layer at level 1.  Bounds as for the
    kitchen / block-push kernels: no less accurate than the block form (x1.5), and within twice the larger error of it."""
    from beso_amd import _lib
    lib = _lib.load()
    cfg = O.CONFIGS["long_horizon"]
    wts = O.make_weights(cfg, seed=3, std=0.02)
    m = make_module(cfg, wts, "bf16")
    s_np, g_np, a_np = O.make_inputs(cfg, B, seed=9 + t, t=t)
    sg_np = np.linspace(0.05, 1.0, B).astype(np.float32)
    s, a, g, sg = G(s_np), G(a_np), G(g_np), G(sg_np)
    ref = O.denoise(wts, cfg, s_np, a_np, g_np, sg_np)
    outs = {}
    try:
        with torch.no_grad():
            for lvl in (2, 1):
                set_level(lvl)
                n = count_fused_launches(lambda: outs.__setitem__(lvl, m(s, a, g, sg).cpu().numpy()))
                assert n == (1 if lvl == 2 else cfg.n_layers), (lvl, n)
    finally:
        set_level(2)
    e_f, e_p, e_fp = rel_err(outs[2], ref), rel_err(outs[1], ref), rel_err(outs[2], outs[1])
    print(f"[parity] long_horizon B={B} t={t}: one-launch-vs-oracle {e_f:.3e} block-kernels-vs-oracle {e_p:.3e} one-launch-vs-blocks {e_fp:.3e}")
    assert e_f < 4e-3 and e_f < 1.5 * e_p + 1e-4
    assert e_fp < 2.0 * max(e_f, e_p)


def test_long_sequence_classifier_free_pairs_are_one_launch_too():
    """Round 4: the one-launch long-sequence instance (a sample per workgroup) carries classifier-free pairs as TWO PASSES of
    the sample's workgroup -- the conditional pass leaves its head outputs in LDS, the unconditional pass combines
    (classifier_free_sampler.py:35-49) and, in a sampler loop, applies the step's update.  One launch per call; agrees with the
    oracle's combination and with the block-kernel form of the same library (BESO_PLAN_BLOCKS: one tail block per layer)."""
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    cfg = O.CONFIGS["long_horizon"]
    wts = O.make_weights(cfg, seed=4, std=0.02)
    m = make_module(cfg, wts, "bf16")
    s_np, g_np, a_np = O.make_inputs(cfg, 3, seed=2)
    sg_np = np.array([0.2, 0.5, 0.9], np.float32)
    lam = 1.5
    model = ClassifierFreeSampleModel(m, lam)
    out = {}
    with torch.no_grad():
        n = count_fused_launches(lambda: out.__setitem__(0, model(G(s_np), G(a_np), G(g_np), G(sg_np)).cpu().numpy()))
        set_level(1)
        try:
            nb = count_fused_launches(lambda: out.__setitem__(1, model(G(s_np), G(a_np), G(g_np), G(sg_np)).cpu().numpy()))
        finally:
            set_level(2)
    assert (n, nb) == (1, cfg.n_layers), (n, nb)
    err = rel_err(out[0], O.denoise_cfg(wts, cfg, s_np, a_np, g_np, sg_np, lam))
    err_b = rel_err(out[0], out[1])
    print(f"[parity] long_horizon classifier-free lambda={lam}: one launch vs oracle {err:.3e}, vs the block kernels {err_b:.3e}")
    assert err < 1e-2 and err_b < 1e-2
    # lambda = 1 / 0 short-circuit to the conditional / unconditional forward (no pair)
    with torch.no_grad():
        for lam1, un in ((1.0, False), (0.0, True)):
            y = ClassifierFreeSampleModel(m, lam1)(G(s_np), G(a_np), G(g_np), G(sg_np))
            assert torch.equal(y, m(G(s_np), G(a_np), G(g_np), G(sg_np), uncond=un))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_fused_sampler_loops_vs_reference_vectors(precision):
    """beso_sample (ddim / euler / heun as one enqueue) against the reference's sampler outputs."""
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    fns = {"ddim": ks.sample_ddim, "euler": ks.sample_euler, "heun": ks.sample_heun, "dpmpp_2m": ks.sample_dpmpp_2m,
           "dpm": ks.sample_dpm_2, "dpmpp_2s": ks.sample_dpmpp_2s}
    for fixture, cfg_name in [("kitchen_samplers.npz", "kitchen"), ("block_push_heun_cfg.npz", "block_push"),
                              ("long_horizon_euler.npz", "long_horizon"), ("long_horizon_euler100.npz", "long_horizon")]:
        fx = load_golden(fixture)
        cfg = O.CONFIGS[cfg_name]
        m = make_module(cfg, _weights(fx, cfg), precision)
        lam = float(fx["cond_lambda"])
        model = m if lam < 0 else ClassifierFreeSampleModel(m, lam)
        for key in sorted(k[:-5] for k in fx if k.endswith("::out")):
            n = int(key.split("_")[-2])
            sampler = key[: key.index(f"_{n}_")]
            x_t = G(fx["x_t"])
            keep = x_t.clone()
            out = fns[sampler](model, G(fx["state"]), x_t, G(fx["goal"]), torch.from_numpy(fx[key + "::sigmas"]),
                               disable=True)
            assert torch.equal(x_t, keep), "sampler must not overwrite the caller's x_T"
            err = rel_err(out.cpu().numpy(), fx[key + "::out"])
            print(f"[parity] {fixture}:{key} {precision}: {err:.3e}")
            tol = TOL[precision] * (1 if n <= 10 else 8) if precision == "fp32" else 2e-2      # bf16 loops: measured 2.1e-3 .. 6.4e-3
            assert err < tol, key


@pytest.mark.parametrize("cfg_name,precision,B,lam", [
    ("kitchen", "bf16", 3, 1.0), ("kitchen", "bf16", 64, 1.0), ("kitchen", "bf16", 700, 1.0), ("kitchen", "bf16", 1100, 1.0),
    ("kitchen", "bf16x3", 5, 1.0), ("kitchen", "bf16x3", 600, 1.0),
    ("block_push", "bf16", 130, 2.0), ("block_push", "bf16", 1030, 2.0), ("block_push", "bf16x3", 258, 2.0),
    ("block_push", "bf16", 9, 0.0), ("long_horizon", "bf16", 5, 1.0), ("long_horizon", "bf16", 3, 1.5)])
def test_sampler_loop_is_one_launch_and_equals_the_stepwise_loop(cfg_name, precision, B, lam):
    """K8 fused into K7: beso_sample runs the whole DDIM / Euler / Heun loop inside ONE launch
    of layers_kernel -- the workgroup that owns a sample applies the step's update in the head and feeds itself the next
    evaluation -- and the result equals the step-by-step form (forward launch + update launch per evaluation:
    BESO_SAMPLE_STEPWISE) BIT FOR BIT, in every instance of the kernel (2 / 4 / 8 samples per workgroup, the split-bf16
    instances, classifier-free pairs, the long-sequence instance), for full and short windows.  Loops of more than 128
    evaluations are cut at step boundaries."""
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    cfg = O.CONFIGS[cfg_name]
    m = make_module(cfg, O.make_weights(cfg, seed=3, std=0.03), precision)
    s_np, g_np, x_np = O.make_inputs(cfg, B, seed=11)
    with torch.no_grad():
        for t in sorted({cfg.obs_seq_len, max(1, cfg.obs_seq_len - 2)}):
            s, g, x = G(s_np[:, :t]), G(g_np), G(x_np[:, :t])
            for sampler, n_steps, n_evals in (("ddim", 3, 3), ("euler", 7, 7), ("heun", 4, 7), ("heun", 70, 139)):
                if n_steps == 70 and (B > 200 or t != cfg.obs_seq_len):
                    continue
                sig = ks.get_sigmas_exponential(n_steps, 0.05, 1.0)
                out = {}
                n_loop = count_fused_launches(lambda: out.__setitem__(
                    "loop", m.fused_sampler(sampler, s, x, g, sig, cond_lambda=lam)))
                n_step = count_fused_launches(lambda: out.__setitem__(
                    "step", m.fused_sampler(sampler, s, x, g, sig, cond_lambda=lam, stepwise=True)))
                assert n_loop == (n_evals + 127) // 128, (sampler, n_steps, n_loop)
                assert n_step == n_evals, (sampler, n_steps, n_step)
                assert torch.isfinite(out["loop"]).all()
                assert torch.equal(out["loop"], out["step"]), (cfg_name, precision, B, t, sampler, n_steps)
            # sample_euler_ancestral (gc_sampling.py:216-256; the README's recommended kitchen sampler): Euler to sigma_down,
            # then the caller's noise of the step times sigma_up -- one launch as well (round 4), eta = 1 and eta = 0
            for n_steps, eta in ((5, 1.0), (3, 0.0), (130, 1.0)):
                if n_steps == 130 and (B > 200 or t != cfg.obs_seq_len):
                    continue
                sig = ks.get_sigmas_exponential(n_steps, 0.05, 1.0)
                nz = torch.randn((n_steps,) + tuple(x.shape), device=DEV, generator=torch.Generator(DEV).manual_seed(5))
                out = {}
                n_loop = count_fused_launches(lambda: out.__setitem__(
                    "loop", m.fused_sampler("euler_ancestral", s, x, g, sig, cond_lambda=lam, eta=eta, noise=nz)))
                n_step = count_fused_launches(lambda: out.__setitem__(
                    "step", m.fused_sampler("euler_ancestral", s, x, g, sig, cond_lambda=lam, eta=eta, noise=nz, stepwise=True)))
                assert (n_loop, n_step) == ((n_steps + 127) // 128, n_steps), (n_loop, n_step)
                assert torch.isfinite(out["loop"]).all()
                assert torch.equal(out["loop"], out["step"]), (cfg_name, precision, B, t, "euler_ancestral", n_steps, eta)
                if eta == 1.0 and n_steps == 5:      # the noise matters
                    other = m.fused_sampler("euler_ancestral", s, x, g, sig, cond_lambda=lam, eta=eta, noise=nz * 0.5)
                    assert not torch.equal(other, out["loop"])


def test_stochastic_and_adaptive_samplers_on_gpu():
    """lms, dpm_fast, dpm_adaptive (incl. rejected steps), dpmpp_2s_ancestral and dpmpp_sde with the HIP denoiser
    (fp32 mode) against the reference's outputs; the injected noise is drawn from the seeded CPU generator the
    reference run used and copied to the device."""
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from test_host_logic import _more_sampler_runs
    fx = load_golden("tiny_more_samplers.npz")
    cfg = O.TINY
    m = make_module(cfg, O.make_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"])), "fp32")
    cpu_noise = lambda s, sn: torch.randn(fx["x_t"].shape).to(DEV)      # noqa: E731
    runs = _more_sampler_runs(m, fx, lambda k: G(fx[k]), cpu_noise)
    sig = torch.from_numpy(fx["sigmas"])
    runs["dpmpp_2s_ancestral"] = lambda: ks.sample_dpmpp_2s_ancestral(m, G(fx["state"]), G(fx["x_t"]), G(fx["goal"]), sig,
                                                                      disable=True, noise_sampler=cpu_noise)
    for key in ("ancestral", "dpm_fast_7_eta", "dpm_adaptive_3_eta"):    # these draw with randn_like on the device
        runs.pop(key)
    for key, fn in runs.items():
        torch.manual_seed(999)
        y = fn()
        if isinstance(y, tuple):
            y, info = y
            assert [info[k] for k in ("steps", "nfe", "n_accept", "n_reject")] == list(fx[key + "::info"]), key
        err = rel_err(y.cpu().numpy(), fx[key])
        print(f"[parity] tiny_more_samplers:{key} fp32: {err:.3e}")
        assert err < 5e-5, key


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_classifier_free_guidance(precision):
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    fx = load_golden("block_push_cfg.npz")
    cfg = O.BLOCK_PUSH
    m = make_module(cfg, _weights(fx, cfg), precision)
    s, a, g, sg = (G(fx[k]) for k in ("state", "action", "goal", "sigma"))
    with torch.no_grad():
        for lam in fx["lambdas"]:
            out = ClassifierFreeSampleModel(m, float(lam))(s, a, g, sg)
            assert rel_err(out.cpu().numpy(), fx[f"lam{float(lam)}"]) < TOL[precision], lam


def test_agent_predict_trace_on_gpu():
    """BesoAgent.predict through the EMA packed-image path (no store/copy_to/restore) == reference trace."""
    from test_host_logic import build_agent, run_agent_trace
    fx = load_golden("tiny_agent_trace.npz")
    cfg = O.TINY
    w = weights_from_fixture(fx)
    agent = build_agent(cfg, lambda: make_module(cfg, w, "fp32"), device=DEV)
    agent.ema_helper.load_shadow_params(agent.model.get_params())
    assert run_agent_trace(agent, fx, device=DEV) < 5e-5
    assert agent._ema_packed is not None, "the EMA packed image must have been used"
    # perturb the live weights: predictions must still come from the (unchanged) EMA shadow
    with torch.no_grad():
        for p in agent.model.parameters():
            p.add_(0.05)
    assert run_agent_trace(agent, fx, device=DEV) < 5e-5
    # ... and follow the shadow when it changes
    agent.ema_helper.load_shadow_params(agent.model.get_params())
    assert run_agent_trace(agent, fx, device=DEV) > 1e-3


def test_repack_when_parameters_change():
    cfg = O.TINY
    w = O.make_weights(cfg, seed=5, std=0.05)
    m = make_module(cfg, w, "fp32")
    s, g, a = (G(v) for v in O.make_inputs(cfg, 4, seed=1))
    sg = G(np.full(4, 0.3, np.float32))
    with torch.no_grad():
        y0 = m(s, a, g, sg).clone()
        m.inner_model.tok_emb.weight.mul_(1.5)              # in-place: version counter bumps
        y1 = m(s, a, g, sg)
    w2 = dict(w)
    w2["inner_model.tok_emb.weight"] = w["inner_model.tok_emb.weight"] * 1.5
    assert rel_err(y1.cpu().numpy(), O.denoise(w2, cfg, *(v.cpu().numpy() for v in (s, a, g, sg)))) < TOL["fp32"]
    assert not torch.allclose(y0, y1)


def test_autograd_comparator_equals_hip_forward():
    """The tests' torch-autograd comparator (tests/autograd_reference.py, pinned to the reference in the CPU suite)
    and the HIP forward agree: what the training-step tests compare against is the same function."""
    from autograd_reference import forward_autograd
    cfg = O.KITCHEN
    m = make_module(cfg, O.make_weights(cfg, seed=3, std=0.04), "fp32")
    s, g, a = (G(v) for v in O.make_inputs(cfg, 8, seed=2))
    sg = G(np.exp(np.random.default_rng(0).uniform(np.log(0.005), 0, 8)).astype(np.float32))
    with torch.no_grad():
        hip = m.inner_model(s, a, g, sg)
    ref = forward_autograd(m.inner_model, s, a, g, sg, False)
    assert rel_err(hip.cpu().numpy(), ref.detach().cpu().numpy()) < TOL["fp32"]


# ------------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["bf16", "fp32", "bf16x3"])
def test_full_batch_properties_kitchen_4096(precision):
    """Config 2 (kitchen, B=4096): samples are independent, so (i) any slice of the batch must equal
    the same samples run alone, bit for bit; (ii) a batch of copies returns copies; (iii) a few
    samples checked against the oracle; (iv) CFG is affine in lambda."""
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    cfg = O.KITCHEN
    w = O.make_weights(cfg, seed=0, std=0.02)
    m = make_module(cfg, w, precision)
    B = 4096
    s_np, g_np, a_np = O.make_inputs(cfg, B, seed=0)
    sg_np = np.exp(np.random.default_rng(1).uniform(np.log(0.005), 0.0, B)).astype(np.float32)
    s, g, a, sg = G(s_np), G(g_np), G(a_np), G(sg_np)
    with torch.no_grad():
        full = m(s, a, g, sg)
        assert torch.isfinite(full).all()
        part = m(s[1000:1256], a[1000:1256], g[1000:1256], sg[1000:1256])
        assert torch.equal(full[1000:1256], part), "batch slicing changed the result"
        rep = m(s[:1].expand(512, -1, -1), a[:1].expand(512, -1, -1), g[:1].expand(512, -1, -1), sg[:1].expand(512))
        assert torch.equal(rep, rep[:1].expand_as(rep))
        idx = [0, 1, 2047, 4095]
        ref = O.denoise(w, cfg, s_np[idx], a_np[idx], g_np[idx], sg_np[idx])
        assert rel_err(full[idx].cpu().numpy(), ref) < TOL[precision]
        out_c = m(s, a, g, sg)
        out_u = m(s, a, g, sg, uncond=True)
        out_2 = ClassifierFreeSampleModel(m, 2.0)(s, a, g, sg)
        lin = out_u + 2.0 * (out_c - out_u)
        assert rel_err(out_2.cpu().numpy(), lin.cpu().numpy()) < 1e-5
        assert torch.equal(ClassifierFreeSampleModel(m, 1.0)(s, a, g, sg), out_c)
        assert torch.equal(ClassifierFreeSampleModel(m, 0.0)(s, a, g, sg), out_u)


def test_sampler_properties_block_push_2048():
    """Config 4 shape (block-push, B=2048, Heun + CFG): the last DDIM step returns the denoised action
    exactly; a fused loop equals the same loop driven step by step from Python; fixed points of the
    schedule (sigmas of length 2) reduce to one denoiser call."""
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    cfg = O.BLOCK_PUSH
    # fp32 mode: bf16 rounding makes the network discontinuous at the 1e-3 level, so "same loop, two
    # drivers" is only an equality at fp32 round-off in the exact mode
    m = make_module(cfg, O.make_weights(cfg, seed=7, std=0.05), "fp32")
    model = ClassifierFreeSampleModel(m, 2.0)
    B = 2048
    s, g, x = (G(v) for v in O.make_inputs(cfg, B, seed=3))
    with torch.no_grad():
        one = ks.sample_ddim(model, s, x, g, torch.tensor([0.7, 0.0]), disable=True)
        den = model(s, x, g, torch.full((B,), 0.7, device=DEV))
        assert torch.equal(one, den)
        sig = ks.get_sigmas_exponential(6, 0.05, 1.0)
        fused = ks.sample_heun(model, s, x, g, sig, disable=True)
        stepwise = ks.sample_heun(model, s, x, g, sig, disable=True, callback=lambda info: None)   # generic loop
        assert rel_err(fused.cpu().numpy(), stepwise.cpu().numpy()) < 1e-5
        assert torch.isfinite(fused).all()


def test_block_push_2048_heun50_cfg_bf16():
    """BASELINE configs[3] at full size: block-push, B = 2048, 50-step Heun, classifier-free guidance lambda = 2 (198
    score-net forwards per sample, all enqueued by one beso_sample call) in the bf16 throughput mode.  Checked against
    the same run in the split-bf16 parity mode (itself held to the oracle on spot samples), sample slices against the
    same samples run alone (bit for bit: samples are independent), and for finiteness."""
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    cfg = O.BLOCK_PUSH
    w = O.make_weights(cfg, seed=7, std=0.05)
    B = 2048
    s_np, g_np, x_np = O.make_inputs(cfg, B, seed=3)
    s, g, x = G(s_np), G(g_np), G(x_np)
    sig = ks.get_sigmas_exponential(50, 0.05, 1.0)
    outs = {}
    with torch.no_grad():
        for prec in ("bf16", "bf16x3"):
            model = ClassifierFreeSampleModel(make_module(cfg, w, prec), 2.0)
            n = count_fused_launches(lambda: outs.__setitem__(prec, ks.sample_heun(model, s, x, g, sig, disable=True)))
            assert n == 1, n                                               # 2 * 50 - 1 evaluations (cond + uncond inside) in ONE launch
            assert torch.isfinite(outs[prec]).all()
            if prec == "bf16":
                part = ks.sample_heun(model, s[512:640], x[512:640], g[512:640], sig, disable=True)
                assert torch.equal(outs[prec][512:640], part)
    e_modes = rel_err(outs["bf16"].cpu().numpy(), outs["bf16x3"].cpu().numpy())
    idx = [0, 1, 1023, 2047]
    ref = O.sample_heun(O.make_model(w, cfg, cond_lambda=2.0), s_np[idx], x_np[idx], g_np[idx], sig.numpy())
    e_x3 = rel_err(outs["bf16x3"][idx].cpu().numpy(), ref)
    e_bf = rel_err(outs["bf16"][idx].cpu().numpy(), ref)
    print(f"[parity] block-push B=2048 Heun-50 x CFG: bf16 vs bf16x3 {e_modes:.3e}; vs oracle (4 samples): bf16x3 {e_x3:.3e}, bf16 {e_bf:.3e}")
    assert e_x3 < 3e-4 and e_modes < 2e-2 and e_bf < 2e-2


def test_long_horizon_256_euler100_bf16():
    """BASELINE configs[4] at one GPU's share: long-horizon (D = 512, 67 tokens), B = 256, 100 Euler steps, bf16 -- every
    one of the 100 forwards ONE launch of the long-sequence instance (a sample per workgroup), all enqueued by one
    beso_sample call.  Checked: the launch count; sample slices against the same samples run alone (bit for bit: a sample is
    its own workgroup, whatever the batch); the same run with the block kernels (two launches per layer) to 2e-2; four
    samples against the oracle's Euler loop; finiteness."""
    from beso_amd import _lib
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    lib = _lib.load()
    cfg = O.CONFIGS["long_horizon"]
    w = O.make_weights(cfg, seed=7, std=0.03)
    m = make_module(cfg, w, "bf16")
    B = 256
    s_np, g_np, x_np = O.make_inputs(cfg, B, seed=3)
    s, g, x = G(s_np), G(g_np), G(x_np)
    sig = ks.get_sigmas_exponential(100, 0.05, 1.0)
    outs = {}
    try:
        with torch.no_grad():
            for lvl in (2, 1):
                set_level(lvl)
                n = count_fused_launches(lambda: outs.__setitem__(lvl, ks.sample_euler(m, s, x, g, sig, disable=True)))
                assert n == (1 if lvl == 2 else 100 * cfg.n_layers), (lvl, n)       # the whole loop is one launch
                assert torch.isfinite(outs[lvl]).all()
            set_level(2)
            part = ks.sample_euler(m, s[100:103], x[100:103], g[100:103], sig, disable=True)
            assert torch.equal(outs[2][100:103], part)
    finally:
        set_level(2)
    idx = [0, 1, 127, 255]
    ref = O.sample_euler(O.make_model(w, cfg), s_np[idx], x_np[idx], g_np[idx], sig.numpy())
    e_forms = rel_err(outs[2].cpu().numpy(), outs[1].cpu().numpy())
    e_or = rel_err(outs[2][idx].cpu().numpy(), ref)
    print(f"[parity] long-horizon B=256 Euler-100: one launch vs block kernels {e_forms:.3e}; vs oracle (4 samples) {e_or:.3e}")
    assert e_forms < 2e-2 and e_or < 2e-2


def test_ragged_and_edge_shapes():
    """B = 1 rollouts, t < W warm-up windows, a batch that is not a multiple of any tile, shared goal."""
    cfg = O.KITCHEN
    w = O.make_weights(cfg, seed=9, std=0.03)
    m = make_module(cfg, w, "fp32")
    with torch.no_grad():
        for B, t in [(1, 1), (1, 4), (3, 2), (37, 3), (129, 4)]:
            s_np, g_np, a_np = O.make_inputs(cfg, B, seed=B + t, t=t)
            sg_np = np.linspace(0.01, 1.0, B).astype(np.float32)
            out = m(G(s_np), G(a_np), G(g_np), G(sg_np))
            assert out.shape == (B, t, cfg.act_dim)
            assert rel_err(out.cpu().numpy(), O.denoise(w, cfg, s_np, a_np, g_np, sg_np)) < TOL["fp32"], (B, t)
        # goal given once for the whole batch ([G, obs]) as predict() does (beso_agent.py:328-329)
        s_np, g_np, a_np = O.make_inputs(cfg, 5, seed=77)
        out = m(G(s_np), G(a_np), G(g_np[0]), G(np.full(5, 0.2, np.float32)))
        ref = O.denoise(w, cfg, s_np, a_np, np.broadcast_to(g_np[0], g_np.shape), np.full(5, 0.2, np.float32))
        assert rel_err(out.cpu().numpy(), ref) < TOL["fp32"]
        with pytest.raises(ValueError):
            m(G(np.zeros((2, 5, 30), np.float32)), G(np.zeros((2, 5, 9), np.float32)), G(g_np[:2]), G(np.ones(2, np.float32)))


@pytest.mark.gpu
@pytest.mark.parametrize("T,H,hd", [(17, 4, 64), (33, 2, 32), (67, 8, 64), (96, 4, 48), (128, 2, 64)])
def test_generic_mfma_attention_matches_fp32_attention(T, H, hd):
    """The bf16 MFMA attention kernel of the generic path (16 < T <= 128 tokens, hd <= 64; score_gpts.py:69-76):
    one-layer networks of shapes that no fused kernel takes, so the whole forward runs on the generic
    kernels, against the oracle.  Window chosen so that 1 + G + 2W = T (ragged last tile, full tiles,
    the 128-token maximum)."""
    G_len = 2 if (T - 3) % 2 == 0 else 1
    W = (T - 1 - G_len) // 2
    if 1 + G_len + 2 * W != T:
        pytest.skip("T not representable as 1 + G + 2W")
    cfg = O.ScoreGPTConfig(obs_dim=6, act_dim=3, embed_dim=H * hd, n_layers=1, n_heads=H, goal_seq_len=G_len,
                           obs_seq_len=W, sigma_data=0.5)
    w = O.make_weights(cfg, seed=5, std=0.05)
    s, g, a = O.make_inputs(cfg, 5, seed=3)
    sg = np.linspace(0.1, 0.9, 5).astype(np.float32)
    ref = O.denoise(w, cfg, s, a, g, sg)
    m = make_module(cfg, w, "bf16")
    with torch.no_grad():
        out = m(G(s), G(a), G(g), G(sg)).cpu().numpy()
    err = rel_err(out, ref)
    print(f"[parity] generic MFMA attention T={T} H={H} hd={hd}: {err:.3e}")
    assert err < TOL["bf16"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["adamw", "adam_l2"])
def test_fused_adam_matches_torch(kind):
    """beso_adam_step (all tensors + EMA in one launch) against torch.optim.AdamW / Adam and the eager EMA
    (beso_agent.py:236-244, ema.py:45-53) over several steps with a StepLR schedule."""
    from beso_amd.optim import FusedAdam, maybe_fuse
    from beso_amd.networks.ema_helper.ema import ExponentialMovingAverage
    torch.manual_seed(0)
    shapes = [(360, 30), (360,), (1440, 360), (5000,), (9, 360), (1,)]
    ref_p = [torch.nn.Parameter(torch.randn(s, device=DEV) * 0.1) for s in shapes]
    fus_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    if kind == "adamw":
        ref = torch.optim.AdamW(ref_p, lr=1e-3, betas=(0.9, 0.999))
        fus = maybe_fuse(torch.optim.AdamW(fus_p, lr=1e-3, betas=(0.9, 0.999)))
    else:
        ref = torch.optim.Adam(ref_p, lr=2e-3, weight_decay=0.05)
        fus = maybe_fuse(torch.optim.Adam(fus_p, lr=2e-3, weight_decay=0.05))
    assert isinstance(fus, FusedAdam)
    s_ref = torch.optim.lr_scheduler.StepLR(ref, 2, 0.5)
    s_fus = torch.optim.lr_scheduler.StepLR(fus, 2, 0.5)
    e_ref = ExponentialMovingAverage(ref_p, 0.999, DEV)
    e_fus = ExponentialMovingAverage(fus_p, 0.999, DEV)
    for it in range(6):
        grads = [torch.randn_like(p) for p in ref_p]
        ref.zero_grad(); fus.zero_grad()
        for p, q, g in zip(ref_p, fus_p, grads):
            p.grad = g.clone()
            q.grad = g.clone()
        ref.step(); s_ref.step()
        if it != 3:                                   # one step without an EMA update (NULL shadow pointer path)
            e_ref.update(ref_p)
            fus.step(ema=e_fus)
        else:
            fus.step()
        s_fus.step()
    for p, q in zip(ref_p, fus_p):
        assert rel_err(q.detach().cpu().numpy(), p.detach().cpu().numpy()) < 2e-6
    for a, b in zip(e_ref.shadow_params, e_fus.shadow_params):
        assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 2e-6
    assert e_ref.num_updates == e_fus.num_updates == 5 and e_fus.version == 5


@pytest.mark.gpu
def test_fused_adam_sharded_steps_equal_the_full_step():
    """FusedAdam.step(shard=(lo, hi)) updates exactly the elements [lo, hi) of the flat parameter order (the owned range
    of the sharded data-parallel exchange): stepping the shards of a 3-way split one after the other is the full step,
    bit for bit (parameters and EMA shadow), and leaves everything outside a shard untouched."""
    from beso_amd.optim import FusedAdam
    from beso_amd.networks.ema_helper.ema import ExponentialMovingAverage
    torch.manual_seed(0)
    shapes = [(7, 5), (4100,), (3, 9000), (1,), (64, 33)]

    def make():
        torch.manual_seed(1)
        ps = [torch.nn.Parameter(torch.randn(*sh, device=DEV)) for sh in shapes]
        for p in ps:
            p.grad = torch.randn_like(p)
        return ps, FusedAdam(ps, lr=1e-2, weight_decay=0.01, decoupled_weight_decay=True), ExponentialMovingAverage(ps, 0.999, DEV)

    pf, of, ef = make()
    ps, os_, es = make()
    n = sum(p.numel() for p in pf)
    cuts = [0, 4099, 4099 + 13001, n]
    for it in range(3):
        of.step(ema=ef)
        decay_counted = False
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            before = torch.cat([p.detach().reshape(-1) for p in ps]).clone()
            if decay_counted:                       # one EMA update per step: the warm-up counter advances once
                es.num_updates -= 1
            for st in os_._groups:                  # (the step counter too: three launches, one Adam step)
                if st is not None:
                    st["step"] = it
            os_.step(ema=es, shard=(lo, hi))
            decay_counted = True
            after = torch.cat([p.detach().reshape(-1) for p in ps])
            assert torch.equal(after[:lo], before[:lo]) and torch.equal(after[hi:], before[hi:])
            assert not torch.equal(after[lo:hi], before[lo:hi])
        for a, b in zip(pf, ps):
            assert torch.equal(a, b)
        assert torch.equal(ef._flat, es._flat)


@pytest.mark.gpu
def test_train_step_with_fused_optimizer_matches_eager(monkeypatch):
    """BesoAgent.train_step: fused Adam(W)+EMA launch == eager torch optimizer + EMA helper on the same batches."""
    from test_host_logic import build_agent
    from beso_amd.optim import FusedAdam
    from beso_amd.networks.scaler.scaler_class import Scaler
    cfg = O.TINY
    w = O.make_weights(cfg, seed=2, std=0.05)
    losses, finals, shadows = {}, {}, {}
    import beso_amd.agents.diffusion_agents.beso_agent as agent_mod
    for mode in ("1", "0"):
        if mode == "0":                                   # the eager torch optimizer + EMA helper: keep what Hydra configured
            monkeypatch.setattr(agent_mod, "maybe_fuse", lambda opt: opt)
        agent = build_agent(cfg, lambda: make_module(cfg, w, "fp32"), device=DEV)
        assert isinstance(agent.optimizer, FusedAdam) == (mode == "1")
        agent.get_scaler(Scaler(np.random.default_rng(0).standard_normal((64, cfg.obs_dim)).astype(np.float32),
                                np.random.default_rng(1).standard_normal((64, cfg.act_dim)).astype(np.float32), True, DEV))
        agent.set_bounds(agent.scaler)
        torch.manual_seed(7)
        batch = {"observation": torch.randn(16, cfg.obs_seq_len, cfg.obs_dim, device=DEV),
                 "action": torch.randn(16, cfg.obs_seq_len, cfg.act_dim, device=DEV),
                 "goal_observation": torch.randn(16, cfg.goal_seq_len, cfg.obs_dim, device=DEV)}
        losses[mode] = [agent.train_step(batch) for _ in range(4)]
        finals[mode] = [p.detach().cpu().numpy().copy() for p in agent.model.parameters()]
        shadows[mode] = [s.cpu().numpy().copy() for s in agent.ema_helper.shadow_params]
    assert np.allclose(losses["1"], losses["0"], rtol=1e-5, atol=1e-6), (losses["1"], losses["0"])
    # Adam's update is lr * m / (sqrt(v) + eps): for elements whose gradient is ~0 a last-bit difference moves
    # the parameter by O(lr) (= 1e-4 here); the bar is a few lr relative to the tensor's largest entry
    for a, b in zip(finals["1"], finals["0"]):
        assert rel_err(a, b) < 5e-3
    for a, b in zip(shadows["1"], shadows["0"]):
        assert rel_err(a, b) < 5e-3


@pytest.mark.gpu
def test_graphed_train_step_matches_eager(monkeypatch):
    """train_step with forward + backward replayed as one HIP graph == the eager step, with the random draws
    (noise, sigma) pinned to fixed tensors so that both see the same numbers."""
    from test_host_logic import build_agent
    from beso_amd.networks.scaler.scaler_class import Scaler
    cfg = O.TINY
    w = O.make_weights(cfg, seed=2, std=0.05)
    torch.manual_seed(11)
    B = 16
    noise = torch.randn(B, cfg.obs_seq_len, cfg.act_dim, device=DEV)
    sigma = torch.rand(B, device=DEV) * 0.9 + 0.05
    batches = [{"observation": torch.randn(B, cfg.obs_seq_len, cfg.obs_dim, device=DEV),
                "action": torch.randn(B, cfg.obs_seq_len, cfg.act_dim, device=DEV),
                "goal_observation": torch.randn(B, cfg.goal_seq_len, cfg.obs_dim, device=DEV)} for _ in range(4)]
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BESO_AMD_TRAIN_GRAPH", mode)
        agent = build_agent(cfg, lambda: make_module(cfg, w, "fp32"), device=DEV)
        agent.get_scaler(Scaler(np.random.default_rng(0).standard_normal((64, cfg.obs_dim)).astype(np.float32),
                                np.random.default_rng(1).standard_normal((64, cfg.act_dim)).astype(np.float32), True, DEV))
        agent.set_bounds(agent.scaler)
        monkeypatch.setattr(agent, "make_sample_density", lambda: (lambda shape, device: sigma.clone()))
        monkeypatch.setattr(torch, "randn_like", lambda t, *a, **k: noise.clone())
        losses = [agent.train_step(b) for b in batches]
        monkeypatch.undo()
        assert (len(agent._train_graphs) == 1) == (mode == "1")
        results[mode] = (losses, [p.detach().cpu().numpy().copy() for p in agent.model.parameters()])
    assert np.allclose(results["1"][0], results["0"][0], rtol=1e-5, atol=1e-6), (results["1"][0], results["0"][0])
    for a, b in zip(results["1"][1], results["0"][1]):
        assert rel_err(a, b) < 5e-3            # Adam amplifies last-bit differences to O(lr), see above


@pytest.fixture
def fused_instance(request):
    """Selects the instance of the fused kernel small batches run on: 'latency' (two samples per workgroup, the default
    up to 512 samples) or 'throughput' (eight per workgroup, what larger batches use)."""
    from beso_amd import _lib
    lib = _lib.load()
    set_instances(0 if request.param == "throughput" else 512)
    yield request.param
    set_instances(512)


@pytest.mark.gpu
@pytest.mark.parametrize("fused_instance", ["latency", "throughput"], indirect=True)
@pytest.mark.parametrize("cfg_name", ["kitchen", "block_push"])
def test_fused_path_ragged_shapes_and_cfg(cfg_name, fused_instance):
    """The fused layers kernel (bf16) on what rollouts feed it: B = 1, batches that do not fill a workgroup's
    sample tile (or leave its last tile ragged), every warm-up window t = 1 .. W, unconditional calls, and
    classifier-free pairs with an odd batch -- each against the oracle, through both instances of the kernel."""
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    cfg = O.CONFIGS[cfg_name]
    w = O.make_weights(cfg, seed=21, std=0.03)
    m = make_module(cfg, w, "bf16")
    worst = 0.0
    with torch.no_grad():
        for B, t in [(1, 1), (1, cfg.obs_seq_len), (3, 2), (9, cfg.obs_seq_len - 1), (37, 3), (129, cfg.obs_seq_len)]:
            s_np, g_np, a_np = O.make_inputs(cfg, B, seed=100 * B + t, t=t)
            sg_np = np.linspace(0.06, 1.0, B).astype(np.float32)
            s, a, g, sg = G(s_np), G(a_np), G(g_np), G(sg_np)
            out = m(s, a, g, sg)
            assert out.shape == (B, t, cfg.act_dim)
            worst = max(worst, rel_err(out.cpu().numpy(), O.denoise(w, cfg, s_np, a_np, g_np, sg_np)))
            out_u = m(s, a, g, sg, uncond=True)
            worst = max(worst, rel_err(out_u.cpu().numpy(), O.denoise(w, cfg, s_np, a_np, g_np, sg_np, uncond=True)))
            out_cfg = ClassifierFreeSampleModel(m, 2.0)(s, a, g, sg)
            ref_cfg = O.denoise_cfg(w, cfg, s_np, a_np, g_np, sg_np, 2.0)
            worst = max(worst, rel_err(out_cfg.cpu().numpy(), ref_cfg))
    print(f"[parity] fused ragged/CFG {cfg_name} bf16 ({fused_instance} instance): {worst:.3e}")
    assert worst < TOL["bf16"]


@pytest.mark.parametrize("cfg_name", ["kitchen", "block_push"])
def test_bf16x3_ragged_shapes_and_cfg(cfg_name):
    """The split-bf16 instance on what rollouts feed the kernel (B = 1, ragged batches, every warm-up window, unconditional
    calls, classifier-free pairs with an odd batch) against the oracle at the north-star bound."""
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    cfg = O.CONFIGS[cfg_name]
    w = O.make_weights(cfg, seed=21, std=0.03)
    m = make_module(cfg, w, "bf16x3")
    worst = 0.0
    with torch.no_grad():
        for B, t in [(1, 1), (1, cfg.obs_seq_len), (3, 2), (9, cfg.obs_seq_len - 1), (37, 3), (129, cfg.obs_seq_len)]:
            s_np, g_np, a_np = O.make_inputs(cfg, B, seed=100 * B + t, t=t)
            sg_np = np.linspace(0.06, 1.0, B).astype(np.float32)
            s, a, g, sg = G(s_np), G(a_np), G(g_np), G(sg_np)
            out = m(s, a, g, sg)
            assert out.shape == (B, t, cfg.act_dim)
            worst = max(worst, rel_err(out.cpu().numpy(), O.denoise(w, cfg, s_np, a_np, g_np, sg_np)))
            out_u = m(s, a, g, sg, uncond=True)
            worst = max(worst, rel_err(out_u.cpu().numpy(), O.denoise(w, cfg, s_np, a_np, g_np, sg_np, uncond=True)))
            out_cfg = ClassifierFreeSampleModel(m, 2.0)(s, a, g, sg)
            worst = max(worst, rel_err(out_cfg.cpu().numpy(), O.denoise_cfg(w, cfg, s_np, a_np, g_np, sg_np, 2.0)))
    print(f"[parity] fused ragged/CFG {cfg_name} bf16x3: {worst:.3e}")
    assert worst < TOL["bf16x3"]


@pytest.mark.parametrize("cfg_name", ["kitchen", "block_push"])
def test_fused_instances_are_bit_identical(cfg_name):
    """The latency instances (two samples per workgroup up to 512 samples -- 256 with a classifier-free pair --, four up
    to 1024) and the throughput instance (eight) of the fused kernel run the same per-sample arithmetic: equal bits for every batch size, window, conditioning mode and through a sampler loop.
    (Round 3: the eight-sample kitchen instance gives its EMPTY token slots zero GEMM operands -- an energy measure, fused.hip
    zero_pad_instance -- and the latency instances do not: equal bits here is also the proof that no real token reads them.)"""
    from beso_amd import _lib
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    lib = _lib.load()
    cfg = O.CONFIGS[cfg_name]
    m = make_module(cfg, O.make_weights(cfg, seed=5, std=0.04), "bf16")
    cfgm = ClassifierFreeSampleModel(m, 1.5)
    sig = ks.get_sigmas_exponential(4, 0.05, 1.0)
    try:
        with torch.no_grad():
            for B, t in [(1, 1), (2, cfg.obs_seq_len), (5, 2), (64, cfg.obs_seq_len), (257, cfg.obs_seq_len - 1), (512, cfg.obs_seq_len),
                         (513, cfg.obs_seq_len), (771, 2), (1024, cfg.obs_seq_len)]:      # 2, then 4 samples per workgroup
                s, g, a = (G(v) for v in O.make_inputs(cfg, B, seed=B + t, t=t))
                sg = G(np.linspace(0.05, 1.0, B).astype(np.float32))
                outs = []
                for limit in (0, 512):
                    set_instances(limit)
                    outs.append((m(s, a, g, sg), m(s, a, g, sg, uncond=True), cfgm(s, a, g, sg),
                                 ks.sample_heun(cfgm, s, a, g, sig, disable=True)))
                for x8, x2 in zip(*outs):
                    assert torch.isfinite(x2).all() and torch.equal(x8, x2), (B, t)
    finally:
        set_instances(512)


@pytest.mark.parametrize("cfg_name", ["kitchen", "block_push"])
def test_bf16x3_instances_are_bit_identical_and_stable_from_run_to_run(cfg_name):
    """The split-bf16 mode has two instances of the kernel: two samples in two token tiles per workgroup (batches up to
    512) and four samples in three (larger batches: half the workgroups, half the L2 -> CU weight stream).  Same per-sample
    arithmetic: equal bits for every batch size, window and conditioning mode, through a sampler loop, and from run to
    run (eight repetitions at B = 4096 -- a variant of the four-sample instance once differed between runs)."""
    from beso_amd import _lib
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    lib = _lib.load()
    cfg = O.CONFIGS[cfg_name]
    m = make_module(cfg, O.make_weights(cfg, seed=5, std=0.04), "bf16x3")
    cfgm = ClassifierFreeSampleModel(m, 1.5)
    sig = ks.get_sigmas_exponential(3, 0.05, 1.0)
    try:
        with torch.no_grad():
            for B, t in [(513, cfg.obs_seq_len), (771, 2), (1030, cfg.obs_seq_len - 1), (4096, cfg.obs_seq_len)]:
                s, g, a = (G(v) for v in O.make_inputs(cfg, B, seed=B + t, t=t))
                sg = G(np.linspace(0.05, 1.0, B).astype(np.float32))
                outs = []
                for limit in (1 << 20, 512):          # two samples per workgroup at every size / four above 512
                    set_instances(limit)
                    outs.append((m(s, a, g, sg), m(s, a, g, sg, uncond=True), cfgm(s, a, g, sg),
                                 ks.sample_heun(cfgm, s, a, g, sig, disable=True)))
                for x2, x4 in zip(*outs):
                    assert torch.isfinite(x4).all() and torch.equal(x2, x4), (B, t)
                if B == 4096:
                    for _ in range(8):
                        assert torch.equal(m(s, a, g, sg), outs[1][0]) and torch.equal(cfgm(s, a, g, sg), outs[1][2])
                        assert torch.equal(ks.sample_heun(cfgm, s, a, g, sig, disable=True), outs[1][3])
    finally:
        set_instances(512)


@pytest.mark.gpu
def test_one_launch_kernels_are_stable_from_run_to_run():
    """Every repetition of a forward / sampler loop equals the first one bit for bit (tests/determinism.py, its quick
    set: the long-horizon instance at windows that leave the fifth token tile nearly empty, and the throughput instances
    of both modes at B = 4096).  Round 3: one instance issued a half k-step's 16x16x16 MFMA two instructions behind the
    16x16x32 MFMA that produced its accumulator -- a distance the MI355X does not interlock and the compiler does not
    pad -- and differed between runs."""
    import determinism                  # tests/determinism.py (the stand-alone form runs the full case list)
    bad = determinism.run(determinism.QUICK, reps=12, verbose=False)
    assert not bad, bad


@pytest.mark.gpu
def test_device_prefetcher_on_gpu():
    """Batches arrive on the device, complete and in order, while their staging buffers are being reused."""
    from beso_amd.data import DevicePrefetcher
    batches = [{"observation": torch.full((64, 4, 30), float(i)), "action": torch.full((64, 4, 9), float(-i)), "i": i}
               for i in range(9)]
    seen = []
    for b in DevicePrefetcher(batches, DEV, depth=2):
        assert b["observation"].is_cuda and b["action"].is_cuda
        seen.append((b["i"], float(b["observation"].sum().item()), float(b["action"].sum().item())))
    assert seen == [(i, 64 * 4 * 30 * float(i), 64 * 4 * 9 * float(-i)) for i in range(9)]


@pytest.mark.gpu
def test_no_reads_past_the_end_of_the_inputs():
    """tools/guard_check.py in a subprocess with the caching allocator off: inputs sit at the end of their own
    allocations, so an out-of-bounds read by any kernel on the path is a GPU memory fault (process abort)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTORCH_NO_CUDA_MEMORY_CACHING="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "guard_check.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "no fault" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name,precision", [("kitchen", "bf16"), ("kitchen", "fp32"), ("block_push", "bf16"),
                                                ("long_horizon", "bf16"), ("kitchen", "bf16x3"), ("block_push", "bf16x3")])
def test_no_writes_outside_output_and_workspace(cfg_name, precision):
    """beso_denoise_fwd / beso_sample through the C ABI with the output, the in/out sample and the workspace
    embedded in larger buffers full of sentinel bytes: nothing outside [ptr, ptr + size) may change."""
    import ctypes as C
    from beso_amd import _lib
    cfg = O.CONFIGS[cfg_name]
    m = make_module(cfg, O.make_weights(cfg, seed=4, std=0.03), precision)
    inner = m.inner_model
    rt, packed = inner.runtime(cfg.sigma_data), inner.packed_weights()
    lib = rt.lib
    PAD = 4096
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = [(1, 1), (5, cfg.obs_seq_len), (67, cfg.obs_seq_len)] if cfg_name != "long_horizon" else [(2, cfg.obs_seq_len)]
    for B, t in shapes:
        s_np, g_np, a_np = O.make_inputs(cfg, B, seed=B + 3 * t, t=t)
        s, g, a = G(s_np), G(g_np), G(a_np)
        sg = G(np.linspace(0.1, 0.9, B).astype(np.float32))
        for lam in (1.0, 2.0):
            two = 1 if lam != 1.0 else 0
            ws_bytes = lib.beso_workspace_bytes(C.byref(rt.cfg), B, t, packed.precision, two)
            wsbig = torch.full((ws_bytes + 2 * PAD,), 0xAB, dtype=torch.uint8, device=DEV)
            n_out = B * t * cfg.act_dim
            outbig = torch.full((n_out + 2048,), -777.0, device=DEV)
            xbig = torch.full((n_out + 2048,), -777.0, device=DEV)
            xbig[1024:1024 + n_out] = a.reshape(-1)
            st = lib.beso_denoise_fwd(C.byref(rt.cfg), packed.buf.data_ptr(), packed.precision, s.data_ptr(), a.data_ptr(),
                                      g.data_ptr(), sg.data_ptr(), outbig.data_ptr() + 4096, B, t, 0, lam,
                                      wsbig.data_ptr() + PAD, ws_bytes, stream)
            _lib.check(st, "denoise_fwd")
            sig = (C.c_float * 4)(1.0, 0.3, 0.05, 0.0)
            st = lib.beso_sample(C.byref(rt.cfg), packed.buf.data_ptr(), packed.precision, _lib.SAMPLER_IDS["heun"],
                                 s.data_ptr(), g.data_ptr(), xbig.data_ptr() + 4096, B, t, sig, 4, lam, 0,
                                 wsbig.data_ptr() + PAD, ws_bytes, stream)
            _lib.check(st, "sample")
            st = lib.beso_sample(C.byref(rt.cfg), packed.buf.data_ptr(), packed.precision, _lib.SAMPLER_IDS["heun"],
                                 s.data_ptr(), g.data_ptr(), xbig.data_ptr() + 4096, B, t, sig, 4, lam, _lib.SAMPLE_STEPWISE,
                                 wsbig.data_ptr() + PAD, ws_bytes, stream)
            _lib.check(st, "sample (stepwise)")
            torch.cuda.synchronize()
            assert bool((wsbig[:PAD] == 0xAB).all()) and bool((wsbig[PAD + ws_bytes:] == 0xAB).all()), (B, t, lam, "workspace")
            for big in (outbig, xbig):
                assert bool((big[:1024] == -777.0).all()) and bool((big[1024 + n_out:] == -777.0).all()), (B, t, lam)
                assert torch.isfinite(big[1024:1024 + n_out]).all()


# -------------------------------------------------------------------------------------------------
# training step in HIP (beso_loss_grad): row f1
# -------------------------------------------------------------------------------------------------
def _train_module(cfg, w, precision, attn_pdrop=0.0, resid_pdrop=0.0, embed_pdrop=0.0, goal_drop=0.0):
    from beso_amd.agents.diffusion_agents.k_diffusion.score_gpts import DiffusionGPT
    from beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser
    inner = functools.partial(
        DiffusionGPT, state_dim=cfg.obs_dim, device=DEV, goal_conditioned=cfg.goal_conditioned,
        action_dim=cfg.act_dim, embed_dim=cfg.embed_dim, embed_pdrob=embed_pdrop, attn_pdrop=attn_pdrop,
        resid_pdrop=resid_pdrop, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=cfg.goal_seq_len,
        obs_seq_len=cfg.obs_seq_len, sigma_vocab_size=3, time_embedding_fn=None, goal_drop=goal_drop,
        linear_output=cfg.linear_output, precision=precision)
    m = GCDenoiser(inner, sigma_data=cfg.sigma_data)
    sd = m.state_dict()
    for k, v in w.items():
        sd[k] = torch.from_numpy(v.copy())
    m.load_state_dict(sd)
    return m.to(DEV).train()


def _train_inputs(cfg, B, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    return (r(B, cfg.obs_seq_len, cfg.obs_dim), r(B, cfg.obs_seq_len, cfg.act_dim), r(B, cfg.goal_seq_len, cfg.obs_dim),
            r(B, cfg.obs_seq_len, cfg.act_dim), (torch.rand(B, generator=g) * 0.9 + 0.05).to(DEV))


def _grad_errors(got, ref, floor=1e-4):
    """per tensor ||got - ref|| / max(||ref||, floor): tensors whose exact gradient is zero (key.bias: softmax is
    invariant to a shift of all scores of a row -- what is left is the rounding noise of the summands) are measured
    against `floor` times the largest gradient entry"""
    gmax = max(r.abs().max().item() for r in ref)
    return [((g - r).norm() / max(r.norm().item(), floor * gmax * r.numel() ** 0.5)).item() for g, r in zip(got, ref)]


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_training_gemm_operand_layouts(precision):
    """The one GEMM kernel of the training step with k-contiguous and contraction-major (k-slow) operands --
    bf16 k-slow tiles come out of LDS through ds_read_b64_tr_b16 -- against fp64 matmul on ragged shapes,
    with and without split-K."""
    import ctypes as C
    from beso_amd import _lib
    lib = _lib.load_dev()             # the development build (include/beso_hip_debug.h): the product library has no such entry
    prec = _lib.PRECISIONS[precision]
    dt = torch.float32 if precision == "fp32" else torch.bfloat16
    tol = 3e-6                                            # (bf16 inputs are exact; products and sums are fp32)
    g = torch.Generator(device="cpu").manual_seed(5)
    for aks, bks in ((0, 0), (0, 1), (1, 1)):
        for (M, N, K, S) in ((128, 128, 64, 1), (200, 136, 72, 1), (360, 1440, 1000, 3), (16, 360, 520, 2), (56, 360, 264, 1)):
            A = torch.randn((K, M) if aks else (M, K), generator=g).to(DEV).to(dt)
            B = torch.randn((K, N) if bks else (N, K), generator=g).to(DEV).to(dt)
            Cm = torch.zeros(M, N, device=DEV)
            st = lib.beso_debug_gemm(prec, aks, bks, A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], Cm.data_ptr(), N, M, N,
                                     K, S, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            _lib.check(st, "debug_gemm")
            ref = (A.t() if aks else A).double() @ (B.t() if bks else B).double().t()
            err = ((Cm.double() - ref).abs().max() / ref.abs().max()).item()
            assert err < tol, (precision, aks, bks, M, N, K, S, err)
    if precision == "bf16":
        # round 5: the panel-owning weight-gradient tiles (both operands k-slow): (1, W) = tiles of 128 rows x ALL N <= 128 W
        # columns, (W, 1) = all M <= 128 W rows x 128 columns, W in {2, 3}; with and without row ranges (slabs behind C)
        for (aks, bks, M, N, K, S) in ((1, 3, 1440, 360, 1000, 1), (3, 1, 360, 1440, 1000, 1), (1, 3, 56, 360, 264, 1),
                                        (1, 3, 16, 384, 4100, 3), (3, 1, 384, 136, 520, 2), (1, 2, 960, 240, 777, 1),
                                        (2, 1, 240, 960, 1030, 4), (1, 2, 200, 256, 64, 1), (3, 1, 8, 8, 8, 1),
                                        (1, 3, 1080, 360, 11264, 1)):
            A = torch.randn(K, M, generator=g).to(DEV).to(dt)
            B = torch.randn(K, N, generator=g).to(DEV).to(dt)
            Cm = torch.full((M * N + 8 + S * (M * N + 8),), float("nan"), device=DEV)      # C, then the slabs of the row ranges
            st = lib.beso_debug_gemm(prec, aks, bks, A.data_ptr(), M, B.data_ptr(), N, Cm.data_ptr(), N, M, N, K, S,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
            _lib.check(st, "debug_gemm (panel tiles)")
            ref = A.double().t() @ B.double()
            err = ((Cm[:M * N].view(M, N).double() - ref).abs().max() / ref.abs().max()).item()
            assert err < tol, ("panel", aks, bks, M, N, K, S, err)
        with pytest.raises(ValueError):          # a tile cannot cover 400 columns with W = 3
            _lib.check(lib.beso_debug_gemm(prec, 1, 3, A.data_ptr(), 400, A.data_ptr(), 400, Cm.data_ptr(), 400, 400, 400, 8, 1, None))
    # unsupported layout pair and misaligned leading dimension are refused, not run
    A = torch.zeros(64, 64, device=DEV)
    with pytest.raises(ValueError):
        _lib.check(lib.beso_debug_gemm(1, 1, 0, A.data_ptr(), 64, A.data_ptr(), 64, A.data_ptr(), 64, 64, 64, 64, 1, None))
    with pytest.raises(_lib.BesoHipError):
        _lib.check(lib.beso_debug_gemm(1, 0, 0, A.data_ptr(), 63, A.data_ptr(), 64, A.data_ptr(), 64, 64, 64, 64, 1, None))


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("fixture,cfg_name", [("tiny_loss.npz", "tiny"), ("kitchen_loss.npz", "kitchen"),
                                              ("block_push_loss.npz", "block_push"), ("tiny_mlp_head_loss.npz", "tiny_mlp_head")])
def test_hip_training_step_matches_reference_gradients(fixture, cfg_name, precision):
    """beso_loss_grad against the loss and the per-parameter gradients that the REFERENCE's loss.backward() produced
    (tests/golden/*_loss.npz: norm and first eight entries of every parameter's gradient, at the kitchen and block-push
    shapes too), through GCDenoiser.loss + autograd's backward.  fp32: the per-op kernels, 5e-4 on the norms.  bf16: the
    LIBRARY'S PLAN -- what `bench.py --workload train` times: at the kitchen and block-push shapes the one-launch forward
    (launch site asserted), the transposed-formulation data gradients with their LayerNorm epilogues, the MFMA attention
    backward and the grouped weight gradients -- held DIRECTLY to the reference's numbers at the bf16 training bound of
    2.6e-2 per tensor (norms; the eight stored entries against the per-entry share of that bound) and 2e-3 on the loss."""
    fx = load_golden(fixture)
    cfg = O.CONFIGS[cfg_name]
    m = _train_module(cfg, _weights(fx, cfg), precision)
    T = lambda k: G(fx[k])
    box = [None]
    n_fused = count_fused_launches(lambda: box.__setitem__(0, m.loss(T("state"), T("action"), T("goal"), T("noise"), T("sigma"))))
    loss = box[0]
    assert "ScoreMatchingLoss" in type(loss.grad_fn).__name__          # the HIP step, not the autograd evaluation
    if precision == "bf16":
        assert n_fused == (1 if cfg_name in ("kitchen", "block_push") else 0), n_fused      # all layers of the forward: one launch
    ltol, ntol = (2e-5, 5e-4) if precision == "fp32" else (2e-3, 2.6e-2)
    assert abs(loss.item() - float(fx["loss"])) < ltol * abs(float(fx["loss"]))
    loss.backward()
    gmax = max(float(fx["gnorm::" + n]) for n, _ in m.named_parameters())
    worst_n, worst_s = (0.0, ""), (0.0, "")
    for n, p in m.named_parameters():
        ref_norm = float(fx["gnorm::" + n])
        got8, ref8 = p.grad.reshape(-1)[:8].double().cpu().numpy(), fx["gslice::" + n].astype(np.float64)
        if precision == "fp32":
            assert abs(p.grad.norm().item() - ref_norm) <= ntol * ref_norm + 1e-6 * gmax, n
            np.testing.assert_allclose(got8, ref8, rtol=5e-3, atol=2e-6 * gmax)
            continue
        # bf16.  A tensor whose exact gradient is zero (key.bias: softmax is invariant to a shift of a row's scores) is
        # rounding noise of the summands: measured against 2e-3 x the largest gradient norm, like _grad_errors does
        floor = 2e-3 * gmax
        en = abs(p.grad.norm().item() - ref_norm) / max(ref_norm, floor)
        # the first eight entries: their error against the share of the tensor's error bound that eight of its numel entries
        # carry (errors spread evenly: |diff8| ~ rel * |g| * sqrt(8 / numel)), or against their own size where that is larger
        k = min(8, p.numel())
        share = max(ref_norm, floor) * (k / p.numel()) ** 0.5
        es = float(np.linalg.norm(got8[:k] - ref8[:k])) / max(share, float(np.linalg.norm(ref8[:k])))
        worst_n, worst_s = max(worst_n, (en, n)), max(worst_s, (es, n))
        assert en < ntol, (n, en)
        assert es < 3 * ntol, (n, es)          # (eight entries: a sample of the tensor's error, not its norm -- 3 x)
    if precision == "bf16":
        print(f"[parity] bf16 training step vs the reference's gradients, {cfg_name}: loss "
              f"{abs(loss.item() - float(fx['loss'])) / abs(float(fx['loss'])):.2e}, worst norm {worst_n[0]:.2e} ({worst_n[1]}), "
              f"worst 8-entry slice {worst_s[0]:.2e} ({worst_s[1]}), fused forward launches {n_fused}")


_LONG_WINDOW = O.ScoreGPTConfig(obs_dim=6, act_dim=4, embed_dim=64, n_layers=2, n_heads=4, goal_seq_len=3, obs_seq_len=9,
                                linear_output=True, sigma_data=0.5)       # T = 22 tokens: the general attention kernels


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("cfg_name,B", [("tiny", 5), ("kitchen", 48), ("block_push", 40), ("long_window", 7), ("tiny_mlp_head", 9)])
def test_hip_loss_and_gradients_match_autograd(cfg_name, B, precision, monkeypatch):
    """Loss and every parameter gradient of the HIP training step against torch autograd on the same function
    (itself pinned to the reference: tests/test_host_logic.py).  fp32 mode: 2e-4 per tensor; bf16 mode (bf16 GEMM
    operands and kept activations, fp32 accumulation): 2.6e-2 per tensor and 2e-3 on the loss -- twice the worst the five
    shapes measure on the MI355X (round 3: gradients 8.4e-3 .. 1.28e-2, loss 7.8e-5 .. 8.3e-4; fp32: 3e-6 .. 1.4e-5)."""
    cfg = {"tiny": O.TINY, "kitchen": O.KITCHEN, "block_push": O.BLOCK_PUSH, "long_window": _LONG_WINDOW,
           "tiny_mlp_head": O.TINY_MLP_HEAD}[cfg_name]         # (the last: Linear(D,100) - SiLU - Linear(100,act) action head)
    m = _train_module(cfg, O.make_weights(cfg, seed=3, std=0.06), precision)
    state, action, goal, noise, sigma = _train_inputs(cfg, B, seed=1)
    from autograd_reference import loss_autograd
    ref_loss = loss_autograd(m, state, action, goal, noise.clone(), sigma)
    ref_loss.backward()
    ref = [p.grad.clone() for p in m.parameters()]
    for p in m.parameters():
        p.grad = None
    loss = m.loss(state, action, goal, noise.clone(), sigma)
    assert "ScoreMatchingLoss" in type(loss.grad_fn).__name__
    loss.backward()
    got = [p.grad for p in m.parameters()]
    ltol, gtol = (2e-5, 1e-4) if precision == "fp32" else (2e-3, 2.6e-2)
    assert abs(loss.item() - ref_loss.item()) < ltol * abs(ref_loss.item())
    errs = _grad_errors(got, ref, 1e-4 if precision == "fp32" else 2e-3)
    worst = max(range(len(errs)), key=lambda i: errs[i])
    print(f"[parity] train grads {cfg_name} B={B} {precision}: loss {abs(loss.item() - ref_loss.item()) / abs(ref_loss.item()):.2e}, "
          f"worst gradient {errs[worst]:.2e} ({list(dict(m.named_parameters()))[worst]})")
    assert errs[worst] < gtol, (list(dict(m.named_parameters()))[worst], errs[worst])
    # a second backward through a scaled loss scales the gradients (autograd contract of the custom node); bias and
    # LayerNorm gradients are accumulated with fp32 atomics, so two runs agree to rounding, not bit for bit
    got = [g.clone() for g in got]
    for p in m.parameters():
        p.grad = None
    (3.0 * m.loss(state, action, goal, noise.clone(), sigma)).backward()
    if precision == "fp32":
        errs3 = _grad_errors([p.grad / 3.0 for p in m.parameters()], got)
        assert max(errs3) < 1e-4, max(errs3)


@pytest.mark.gpu
def test_hip_training_dropout_masks():
    """Dropout of the token embeddings, of the attention weights (kitchen: 0.3) and of the proj / MLP outputs
    (block-push: 0.05) inside the HIP step: the counter-based masks are a function of the seed only (same seed -> same loss and gradients,
    another seed -> another loss), the backward uses the forward's masks (directional derivative by central
    differences at a fixed seed), and the expected loss is near the dropout-free one."""
    from beso_amd.training import HipTrainStep
    cfg = O.TINY
    w = O.make_weights(cfg, seed=4, std=0.08)
    m = _train_module(cfg, w, "fp32", attn_pdrop=0.3, resid_pdrop=0.1, embed_pdrop=0.1)
    inner = m.inner_model
    step = HipTrainStep(inner, cfg.sigma_data)
    state, action, goal, noise, sigma = _train_inputs(cfg, 64, seed=2)
    run = lambda seed: step.run(state, action, goal, noise, sigma, seed=seed, fresh_grads=True)
    l1, f1, _ = run(11)
    l2, f2, _ = run(11)
    l3, _, _ = run(12)
    assert abs(l1.item() - l2.item()) < 1e-6 * abs(l1.item())
    assert (f1 - f2).abs().max().item() < 1e-5 * f1.abs().max().item()     # (atomics: rounding-level differences)
    assert l3.item() != l1.item()
    # eval mode switches the dropouts off
    inner.eval()
    l_eval, _, _ = run(11)
    m0 = _train_module(cfg, w, "fp32")
    l_ref, _, _ = HipTrainStep(m0.inner_model, cfg.sigma_data).run(state, action, goal, noise, sigma, seed=1, fresh_grads=True)
    assert abs(l_eval.item() - l_ref.item()) < 1e-6 * abs(l_ref.item())
    inner.train()
    mean_drop = float(np.mean([run(s)[0].item() for s in range(20, 36)]))
    assert abs(mean_drop - l_ref.item()) < 0.25 * abs(l_ref.item())
    # directional derivative at seed 11
    params = list(inner.parameters())
    gen = torch.Generator(device="cpu").manual_seed(0)
    dirs = [torch.randn(p.shape, generator=gen).to(DEV) * p.detach().abs().mean().clamp_min(1e-3) for p in params]
    _, flat, views = run(11)
    analytic = sum((g.double() * d.double()).sum().item() for g, d in zip(views, dirs))
    eps = 2e-2
    with torch.no_grad():
        for p, d in zip(params, dirs):
            p.add_(eps * d)
        lp = run(11)[0].item()
        for p, d in zip(params, dirs):
            p.sub_(2 * eps * d)
        lm = run(11)[0].item()
        for p, d in zip(params, dirs):
            p.add_(eps * d)
    numeric = (lp - lm) / (2 * eps)
    assert abs(numeric - analytic) < 3e-2 * abs(analytic) + 1e-5, (numeric, analytic)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name,B,precision", [("kitchen", 64, "fp32"), ("block_push", 48, "fp32"), ("kitchen", 64, "bf16")])
def test_hip_training_goal_masking(cfg_name, B, precision):
    """DiffusionGPT.mask_cond in training mode (score_gpts.py:298-299, 360-371; BASELINE configs 3 / 4: cond_mask_prob =
    0.1) inside the HIP step: goals are zeroed ELEMENTWISE over [B, G, obs] with probability goal_drop, kept elements are
    not rescaled.  The kernel's mask for (goal_drop, seed) is read back through beso_goal_mask and injected into the
    torch-autograd comparator: loss and every parameter gradient must agree (fp32 1e-4 / bf16 2.6e-2 per tensor: twice the 1.26e-2 measured)."""
    from autograd_reference import loss_autograd
    cfg = O.CONFIGS[cfg_name]
    m = _train_module(cfg, O.make_weights(cfg, seed=3, std=0.06), precision, goal_drop=0.1)
    inner = m.inner_model
    assert inner.cond_mask_prob == 0.1 and inner.training
    state, action, goal, noise, sigma = _train_inputs(cfg, B, seed=4)
    step = m.hip_train_step(state, action, goal, noise, sigma)
    assert step is not None
    seed = 20240
    mask = step.goal_mask(B, seed)
    assert mask.shape == goal.shape and bool(((mask == 0) | (mask == 1)).all())
    frac = 1.0 - mask.mean().item()
    n = mask.numel()
    assert abs(frac - 0.1) < 4 * (0.09 / n) ** 0.5 + 1e-3, frac                    # Bernoulli(0.1) over B*G*obs elements
    per_sample = mask.reshape(B, -1)
    assert bool(((per_sample.min(1).values == 0) & (per_sample.max(1).values == 1)).any())   # elementwise, not per sample
    assert not torch.equal(mask, step.goal_mask(B, seed + 1)) and torch.equal(mask, step.goal_mask(B, seed))
    assert torch.equal(step.goal_mask(B, seed, goal_drop=0.0), torch.ones_like(mask))
    loss, flat, views = step.run(state, action, goal, noise, sigma, seed=seed, fresh_grads=True)
    got = [v.clone() for v in views]
    # the comparator with the SAME mask injected (its own mask_cond switched off)
    inner.cond_mask_prob = 0.0
    try:
        ref_loss = loss_autograd(m, state, action, goal * mask, noise.clone(), sigma)
        ref_loss.backward()
        ref = [p.grad.clone() for p in m.parameters()]
        # ... and goal_drop = 0 on the pre-masked goals is the same step
        loss0, _, views0 = step.run(state, action, goal * mask, noise, sigma, seed=seed, fresh_grads=True, goal_drop=0.0)
    finally:
        inner.cond_mask_prob = 0.1
        for p in m.parameters():
            p.grad = None
    ltol, gtol, floor = (2e-5, 1e-4, 1e-4) if precision == "fp32" else (2e-3, 2.6e-2, 2e-3)
    assert abs(loss.item() - ref_loss.item()) < ltol * abs(ref_loss.item())
    errs = _grad_errors(got, ref, floor)
    worst = max(range(len(errs)), key=lambda i: errs[i])
    print(f"[parity] goal masking {cfg_name} {precision}: masked fraction {frac:.4f}, loss rel err "
          f"{abs(loss.item() - ref_loss.item()) / abs(ref_loss.item()):.2e}, worst gradient {errs[worst]:.2e}")
    assert errs[worst] < gtol, (list(dict(m.named_parameters()))[worst], errs[worst])
    assert abs(loss0.item() - loss.item()) < 1e-6 * abs(loss.item())
    assert max(_grad_errors([v for v in views0], got, floor)) < 1e-5
    # the unmasked step is a different one
    loss_plain, _, _ = step.run(state, action, goal, noise, sigma, seed=seed, fresh_grads=True, goal_drop=0.0)
    assert abs(loss_plain.item() - loss.item()) > 1e-6 * abs(loss.item())
    # eval mode: no masking whatever cond_mask_prob says (mask_cond is training-only, :364)
    m.eval()
    loss_eval, _, _ = step.run(state, action, goal, noise, sigma, seed=seed, fresh_grads=True)
    m.train()
    assert abs(loss_eval.item() - loss_plain.item()) < 1e-6 * abs(loss_plain.item())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name,B,t", [("kitchen", 200, None), ("kitchen", 37, 2), ("block_push", 96, None)])
def test_training_forward_tail_block_equals_the_per_op_forward(cfg_name, B, t):
    """bf16 training step: the forward runs each layer's out-projection .. next layer's q/k/v as one tile kernel
    (train_tail_kernel, fused.hip) instead of six per-op launches.  Same kept activations in the same formats, same
    arithmetic type; only the accumulation order inside the GEMMs differs -- so loss and gradients of the two forms of
    the forward must agree inside the bf16 bound that holds them to autograd (2.6e-2): here 2e-2 per tensor (measured
    ~5e-3), loss 1e-3; ragged token counts (M not a multiple of the 96-token tile) and a short window included."""
    from beso_amd import _lib
    lib = _lib.load()
    cfg = O.CONFIGS[cfg_name]
    m = _train_module(cfg, O.make_weights(cfg, seed=3, std=0.06), "bf16", attn_pdrop=0.3)
    state, action, goal, noise, sigma = _train_inputs(cfg, B, seed=5)
    if t is not None:
        state, action, noise = state[:, :t].contiguous(), action[:, :t].contiguous(), noise[:, :t].contiguous()
    step = m.hip_train_step(state, action, goal, noise, sigma)
    out = {}
    try:
        for on in (2, 0):                                    # 2: the tile kernel whatever the size; 0: per-op kernels
            set_train_tail(on)
            loss, flat, views = step.run(state, action, goal, noise, sigma, seed=77, fresh_grads=True)
            out[1 if on else 0] = (loss.item(), [v.clone() for v in views])
    finally:
        set_train_tail(1)
    errs = _grad_errors(out[1][1], out[0][1], 2e-3)
    worst = max(range(len(errs)), key=lambda i: errs[i])
    print(f"[parity] tail-block vs per-op training forward {cfg_name} B={B}: loss {abs(out[1][0] - out[0][0]) / abs(out[0][0]):.2e}, "
          f"worst gradient {errs[worst]:.2e} ({list(dict(m.named_parameters()))[worst]})")
    assert abs(out[1][0] - out[0][0]) < 1e-3 * abs(out[0][0])
    assert errs[worst] < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name,B,t", [("kitchen", 200, None), ("kitchen", 37, 2), ("kitchen", 1030, None),
                                          ("kitchen", 3, 1), ("block_push", 96, None), ("block_push", 1100, 3),
                                          ("kitchen", 2300, None)])      # (25,300 token rows: the weight gradients in two row windows)
def test_training_forward_as_one_launch_equals_the_per_op_forward(cfg_name, B, t):
    """bf16 training step, round 4: ALL layers of the forward as ONE launch (train_fwd_kernel, fused.hip: the phases of the
    inference kernel with store hooks for everything the backward keeps, attention dropout inside the core with the per-op
    kernels' mask, the last layer's tail on the compact action rows) -- the library's default where the shape has it.  Same
    kept activations in the same formats and the same dropout mask as the per-op forward; what differs is the accumulation
    order inside the GEMMs and the bf16 MFMA attention core against the per-op fp32 one.  Loss and every gradient of the two
    forms must agree inside the bf16 bound that holds them to autograd (2.6e-2): 2e-2 per tensor, loss 2e-3.  Both instances
    (four samples per workgroup up to 1024 samples, eight beyond), ragged last workgroups and short windows included."""
    cfg = O.CONFIGS[cfg_name]
    m = _train_module(cfg, O.make_weights(cfg, seed=3, std=0.06), "bf16", attn_pdrop=0.3)
    state, action, goal, noise, sigma = _train_inputs(cfg, B, seed=5)
    if t is not None:
        state, action, noise = state[:, :t].contiguous(), action[:, :t].contiguous(), noise[:, :t].contiguous()
    step = m.hip_train_step(state, action, goal, noise, sigma)
    out = {}
    try:
        for on in (1, 0):                                    # 1: the library's choice = one launch; 0: per-op kernels
            set_train_tail(on)
            n = count_fused_launches(lambda: out.__setitem__(on, step.run(state, action, goal, noise, sigma, seed=77, fresh_grads=True)))
            out[on] = (out[on][0].item(), [v.clone() for v in out[on][2]], n)
    finally:
        set_train_tail(1)
    assert (out[1][2], out[0][2]) == (1, 0), (out[1][2], out[0][2])      # launches at the fused kernel's site
    errs = _grad_errors(out[1][1], out[0][1], 2e-3)
    worst = max(range(len(errs)), key=lambda i: errs[i])
    print(f"[parity] one-launch vs per-op training forward {cfg_name} B={B} t={t}: loss {abs(out[1][0] - out[0][0]) / abs(out[0][0]):.2e}, "
          f"worst gradient {errs[worst]:.2e} ({list(dict(m.named_parameters()))[worst]})")
    assert abs(out[1][0] - out[0][0]) < 2e-3 * abs(out[0][0])
    assert errs[worst] < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name,B,resid_p,embed_p", [("kitchen", 100, 0.1, 0.0), ("kitchen", 37, 0.1, 0.1), ("block_push", 96, 0.05, 0.0),
                                                       ("block_push", 50, 0.0, 0.1), ("block_push", 1100, 0.05, 0.0),
                                                       ("kitchen", 1030, 0.1, 0.0), ("block_push", 3, 0.05, 0.0)])
def test_training_backward_kernels_with_residual_and_embedding_dropout_equal_the_per_op_step(cfg_name, B, resid_p, embed_p):
    """bf16 training step with dropout on the residual branches / the embedding (block-push ships resid_pdrop = 0.05): the
    library's backward is the transposed-formulation data-gradient kernels (train_dgrad_kernel, train_mlp_bwd_kernel) with the
    LayerNorm backward as their epilogue, which evaluates the site's mask -- the stand-alone kernel's hash of (seed, site,
    row D + feature); the sigma token's embedding row carries none -- on its own elements.  Same seed = same masks in both plans,
    so the library's choice and the per-op kernels must agree inside the bf16 bound (2e-2 per tensor, loss 2e-3).
    Round 5: the library's FORWARD is the one-launch kernel with resid_pdrop > 0 too (train_fwd_kernel<..., RD = 1>: the branch
    accumulated alone, masked with the per-op forward's hash, the residual added back from the kept x) -- one launch at the
    fused kernel's site is asserted, both instances (four / eight samples per workgroup) and a ragged three-sample batch."""
    cfg = O.CONFIGS[cfg_name]
    m = _train_module(cfg, O.make_weights(cfg, seed=3, std=0.06), "bf16", attn_pdrop=0.3, resid_pdrop=resid_p, embed_pdrop=embed_p)
    m.train()
    state, action, goal, noise, sigma = _train_inputs(cfg, B, seed=5)
    step = m.hip_train_step(state, action, goal, noise, sigma)
    out = {}
    try:
        for on in (1, 0):
            set_train_tail(on)
            n = count_fused_launches(lambda: out.__setitem__(on, step.run(state, action, goal, noise, sigma, seed=77, fresh_grads=True)))
            r = out[on]
            out[on] = (r[0].item(), [v.clone() for v in r[2]], n)
    finally:
        set_train_tail(1)
    assert (out[1][2], out[0][2]) == (1, 0), (out[1][2], out[0][2])      # the forward: one launch / the per-op kernels
    errs = _grad_errors(out[1][1], out[0][1], 2e-3)
    worst = max(range(len(errs)), key=lambda i: errs[i])
    print(f"[parity] library vs per-op training step {cfg_name} B={B} resid_p={resid_p} embed_p={embed_p}: "
          f"loss {abs(out[1][0] - out[0][0]) / abs(out[0][0]):.2e}, worst gradient {errs[worst]:.2e} "
          f"({list(dict(m.named_parameters()))[worst]})")
    assert abs(out[1][0] - out[0][0]) < 2e-3 * abs(out[0][0])
    assert errs[worst] < 2e-2


@pytest.mark.gpu
def test_bf16_training_step_on_random_shapes_stays_as_close_to_fp32_as_the_per_op_plan():
    """tools/fuzz_train_bf16.py for ten seconds: random model shapes, batch sizes, windows and dropouts (the shipped shapes among
    them) through the bf16 HIP step in the library's plan and in the per-op plan, each measured against the fp32 HIP step with
    the same dropout masks -- the library's kernels (matrix-pipe attention backward, transposed data gradients, split / arranged
    weight gradients) are never further from fp32 than the bf16 bound or 1.5 x the per-op plan's own distance."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_train_bf16", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_train_bf16.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, worst, ratio = mod.run(10.0, 21)
    assert n >= 20


@pytest.mark.gpu
def test_agent_train_step_with_goal_drop_runs_the_hip_step():
    """BesoAgent.train_step on a model built with goal_drop = 0.1 and the kitchen dropouts (configs[2] / [3]): the HIP step
    serves it (no host-side masking, no torch-op network), losses are finite and decrease over a few steps."""
    from test_host_logic import build_agent
    from beso_amd.networks.scaler.scaler_class import Scaler
    cfg = O.KITCHEN
    w = O.make_weights(cfg, seed=5, std=0.02)
    agent = build_agent(cfg, lambda: _train_module(cfg, w, "bf16", attn_pdrop=0.3, goal_drop=0.1), device=DEV, lr=1e-3)
    rng = np.random.default_rng(0)
    agent.get_scaler(Scaler(rng.standard_normal((64, cfg.obs_dim)).astype(np.float32),
                            rng.standard_normal((64, cfg.act_dim)).astype(np.float32), True, DEV))
    agent.set_bounds(agent.scaler)
    torch.manual_seed(3)
    batch = {"observation": torch.randn(256, cfg.obs_seq_len, cfg.obs_dim, device=DEV),
             "action": torch.tanh(torch.randn(256, cfg.obs_seq_len, cfg.act_dim, device=DEV)),
             "goal_observation": torch.randn(256, cfg.goal_seq_len, cfg.obs_dim, device=DEV)}
    losses = [agent.train_step(batch) for _ in range(30)]
    assert getattr(agent, "_hip_step", None) is not None
    assert all(np.isfinite(l) for l in losses) and np.mean(losses[-5:]) < 0.9 * np.mean(losses[:5]), losses


@pytest.mark.gpu
def test_rand_log_logistic_as_one_launch_equals_the_elementwise_chain():
    """rand_log_logistic (k_diffusion/utils.py:178-185) on a HIP device: torch.rand in float64, then ONE launch
    (beso_log_logistic) for logit * scale + loc -> exp -> fp32 instead of seven elementwise ones.  Same draws (the generator
    is torch's), same float64 operations: equal to the torch chain to fp32 rounding, inside [min_value, max_value], and the
    quantiles of the shipped density (loc = log 0.5, scale = 0.5, [0.005, 1]) as the reference's."""
    import math
    from beso_amd.agents.diffusion_agents.k_diffusion import utils
    for loc, scale, lo, hi in ((math.log(0.5), 0.5, 0.005, 1.0), (0.0, 1.0, 0.0, float("inf")), (-1.2, 0.3, 0.05, 80.0)):
        torch.manual_seed(17)
        got = utils.rand_log_logistic((4099,), loc=loc, scale=scale, min_value=lo, max_value=hi, device=DEV)
        torch.manual_seed(17)
        u = torch.rand((4099,), device=DEV, dtype=torch.float64)
        cdf = lambda v: 0.0 if v <= 0 else (1.0 if v == float("inf") else 1.0 / (1.0 + math.exp(-(math.log(v) - loc) / scale)))      # noqa: E731
        ref = (u * (cdf(hi) - cdf(lo)) + cdf(lo)).logit().mul(scale).add(loc).exp().to(torch.float32)
        assert got.dtype == torch.float32 and got.shape == ref.shape
        rel = ((got - ref).abs() / ref.abs()).max().item()
        assert rel < 1e-6, (loc, scale, rel)
        assert got.min().item() >= lo * (1 - 1e-6) and got.max().item() <= hi * (1 + 1e-6)
    # CPU tensors keep the torch chain (no HIP call)
    assert utils.rand_log_logistic((5,), loc=0.0, scale=1.0, device="cpu").device.type == "cpu"


@pytest.mark.gpu
def test_train_step_reads_the_loss_on_the_loss_stream(monkeypatch):
    """BesoAgent.train_step returns `loss.item()` (beso_agent.py:248).  The loss is final at the end of the forward half, so
    the step releases a side stream there (beso_loss_grad_streams) and reads the loss on it: the call returns while backward
    and optimizer still run.  Same losses, same parameters as the plain read (BESO_AMD_ASYNC_LOSS=0) over several steps --
    to the rounding of the few atomically accumulated bias gradients --, and the parameters the NEXT step reads are the
    updated ones (the compute stream's order is untouched)."""
    from test_host_logic import build_agent
    from beso_amd.networks.scaler.scaler_class import Scaler
    cfg = O.KITCHEN
    w = O.make_weights(cfg, seed=5, std=0.02)
    rng = np.random.default_rng(0)
    sc = (rng.standard_normal((64, cfg.obs_dim)).astype(np.float32), rng.standard_normal((64, cfg.act_dim)).astype(np.float32))
    torch.manual_seed(3)
    batch = {"observation": torch.randn(256, cfg.obs_seq_len, cfg.obs_dim, device=DEV),
             "action": torch.tanh(torch.randn(256, cfg.obs_seq_len, cfg.act_dim, device=DEV)),
             "goal_observation": torch.randn(256, cfg.goal_seq_len, cfg.obs_dim, device=DEV)}
    runs = {}
    for tag in ("1", "0", "1 again"):
        mode = tag[0]
        monkeypatch.setenv("BESO_AMD_ASYNC_LOSS", mode)
        agent = build_agent(cfg, lambda: _train_module(cfg, w, "bf16", attn_pdrop=0.3, goal_drop=0.1), device=DEV, lr=1e-3)
        agent.get_scaler(Scaler(sc[0], sc[1], True, DEV))
        agent.set_bounds(agent.scaler)
        torch.manual_seed(11)
        losses = [agent.train_step(batch) for _ in range(8)]
        torch.cuda.synchronize()
        assert (getattr(agent, "_loss_ready", None) is not None) == (mode == "1")
        runs[tag] = (np.array(losses), torch.cat([p.detach().reshape(-1) for p in agent.model.parameters()]))
    # (Adam turns the rounding of the atomically accumulated bias gradients into +-lr steps on near-zero gradient elements, so
    # two runs of ONE mode already differ in the parameters: the other mode must not differ by more than that)
    dev = lambda a, b: ((runs[a][1] - runs[b][1]).norm() / runs[b][1].norm()).item()
    ldev = lambda a, b: np.abs(runs[a][0] - runs[b][0]).max() / np.abs(runs[b][0]).max()
    print(f"[parity] loss stream vs plain read: losses {ldev('1', '0'):.2e} (run to run {ldev('1', '1 again'):.2e}), "
          f"parameters {dev('1', '0'):.2e} (run to run {dev('1', '1 again'):.2e})")
    assert np.all(np.isfinite(runs["1"][0])) and ldev("1", "0") < 2e-4
    assert dev("1", "0") < 3 * dev("1", "1 again") + 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_training_step_kitchen_1024_is_the_mean_of_its_slices(precision):
    """BASELINE config 3 per-GPU size (kitchen, 1024 samples per step): the loss is the mean over samples
    (score_wrappers.py:79), so loss and gradients of the whole batch must equal the mean of those of its four 256-sample
    slices (also what makes the data-parallel average of per-rank gradients the global gradient); plus a spot check of
    the first 48 samples' step against torch autograd."""
    from autograd_reference import loss_autograd
    cfg = O.KITCHEN
    m = _train_module(cfg, O.make_weights(cfg, seed=8, std=0.04), precision)
    state, action, goal, noise, sigma = _train_inputs(cfg, 1024, seed=2)
    step = m.hip_train_step(state, action, goal, noise, sigma)
    loss, flat, _ = step.run(state, action, goal, noise, sigma, seed=1, fresh_grads=True)
    acc, lsum = torch.zeros_like(flat), 0.0
    for k in range(4):
        sl = slice(256 * k, 256 * (k + 1))
        l_k, f_k, _ = step.run(state[sl], action[sl], goal[sl], noise[sl], sigma[sl], seed=1, fresh_grads=True)
        acc += f_k / 4
        lsum += l_k.item() / 4
    rel = ((flat - acc).norm() / flat.norm()).item()
    print(f"[parity] kitchen 1024-sample step vs mean of 4 slices ({precision}): loss {abs(loss.item() - lsum) / lsum:.2e}, "
          f"gradient {rel:.2e}")
    ltol, gtol = (1e-5, 1e-4) if precision == "fp32" else (1e-4, 2e-2)
    assert abs(loss.item() - lsum) < ltol * lsum and rel < gtol
    if precision == "fp32":
        sl = slice(0, 48)
        l_s, _, v_s = step.run(state[sl], action[sl], goal[sl], noise[sl], sigma[sl], seed=1, fresh_grads=True)
        got = [v.clone() for v in v_s]
        ref_loss = loss_autograd(m, state[sl], action[sl], goal[sl], noise[sl].clone(), sigma[sl])
        ref_loss.backward()
        ref = [p.grad.clone() for p in m.parameters()]
        assert abs(l_s.item() - ref_loss.item()) < 2e-5 * abs(ref_loss.item())
        assert max(_grad_errors(got, ref)) < 2e-4


@pytest.mark.gpu
def test_inference_on_live_weights_follows_the_fused_optimizer():
    """use_ema = False: evaluate() / model() read the LIVE parameters, which the fused Adam launch updates through raw
    pointers -- the packed-weight cache (keyed on the parameters' version counters) must notice.  After N train_steps the
    HIP forward equals the oracle on the parameters' current values, and differs from the initial ones."""
    from test_host_logic import build_agent
    from beso_amd.optim import FusedAdam
    from beso_amd.networks.scaler.scaler_class import Scaler
    cfg = O.TINY
    w = O.make_weights(cfg, seed=2, std=0.05)
    agent = build_agent(cfg, lambda: make_module(cfg, w, "fp32"), device=DEV, lr=5e-3)
    agent.use_ema = False
    assert isinstance(agent.optimizer, FusedAdam)
    rng = np.random.default_rng(0)
    agent.get_scaler(Scaler(rng.standard_normal((64, cfg.obs_dim)).astype(np.float32),
                            rng.standard_normal((64, cfg.act_dim)).astype(np.float32), True, DEV))
    agent.set_bounds(agent.scaler)
    s_np, g_np, a_np = O.make_inputs(cfg, 8, seed=1)
    sg_np = np.linspace(0.1, 0.9, 8).astype(np.float32)
    agent.model.eval()
    with torch.no_grad():
        y0 = agent.model(G(s_np), G(a_np), G(g_np), G(sg_np)).cpu().numpy()       # packs the initial weights
    torch.manual_seed(7)
    batch = {"observation": torch.randn(16, cfg.obs_seq_len, cfg.obs_dim, device=DEV),
             "action": torch.randn(16, cfg.obs_seq_len, cfg.act_dim, device=DEV),
             "goal_observation": torch.randn(16, cfg.goal_seq_len, cfg.obs_dim, device=DEV)}
    for _ in range(5):
        agent.train_step(batch)
    agent.model.eval()
    now = {k: v.detach().cpu().numpy() for k, v in agent.model.state_dict().items() if not k.endswith("attn.mask")}
    with torch.no_grad():
        y1 = agent.model(G(s_np), G(a_np), G(g_np), G(sg_np)).cpu().numpy()
    assert rel_err(y1, O.denoise(now, cfg, s_np, a_np, g_np, sg_np)) < TOL["fp32"]
    assert rel_err(y1, y0) > 1e-3, "the forward still ran on the first-packed weights"
    mse0 = agent.evaluate(batch)
    for _ in range(5):
        agent.train_step(batch)
    assert agent.evaluate(batch) != mse0


@pytest.mark.gpu
def test_schedules_with_interior_zeros_take_the_stepwise_loop():
    """get_sigmas_linear(sigma_min = 0) ends in [.., 0, 0]: beso_sample rejects non-positive interior values
    (BESO_ERR_BAD_ARG), so the Python samplers must keep such schedules on the step-by-step loop (where the reference's
    arithmetic handles them) instead of raising."""
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    cfg = O.TINY
    w = O.make_weights(cfg, seed=2, std=0.05)
    m = make_module(cfg, w, "fp32")
    s_np, g_np, x_np = O.make_inputs(cfg, 4, seed=1)
    sig = ks.get_sigmas_linear(5, 0.0, 1.0)
    assert float(sig[-2]) == 0.0 and float(sig[-1]) == 0.0
    with torch.no_grad():
        out = ks.sample_ddim(m, G(s_np), G(x_np), G(g_np), sig[:-1], disable=True)           # [1 .. 0]: plain fused call
        ref = O.sample_ddim(O.make_model(w, cfg), s_np, x_np, g_np, sig[:-1].numpy())
        assert rel_err(out.cpu().numpy(), ref) < TOL["fp32"]
        good = ks.get_sigmas_linear(5, 0.01, 1.0)
        fused = ks.sample_euler(m, G(s_np), G(x_np), G(g_np), good, disable=True)
        step = ks.sample_euler(m, G(s_np), G(x_np), G(g_np), good, disable=True, callback=lambda info: None)
        assert rel_err(fused.cpu().numpy(), step.cpu().numpy()) < 1e-5
        # interior zero: the fused entry would raise ValueError; the sampler falls back to the loop, which (like the
        # reference's, gc_sampling.py:205-210) divides by sigma = 0 on the last step -- no exception either way
        out0 = ks.sample_euler(m, G(s_np), G(x_np), G(g_np), sig, disable=True)
        assert out0.shape == (4, cfg.obs_seq_len, cfg.act_dim)


@pytest.mark.gpu
def test_train_step_runs_on_the_hip_step_and_matches_autograd(monkeypatch):
    """BesoAgent.train_step: forward + backward through beso_loss_grad (gradients as views of one flat buffer that
    the fused optimizer reads) == the same steps through torch autograd, with noise and sigma pinned."""
    from test_host_logic import build_agent
    from beso_amd.networks.scaler.scaler_class import Scaler
    cfg = O.TINY
    w = O.make_weights(cfg, seed=2, std=0.05)
    torch.manual_seed(11)
    B = 16
    noise = torch.randn(B, cfg.obs_seq_len, cfg.act_dim, device=DEV)
    sigma = torch.rand(B, device=DEV) * 0.9 + 0.05
    batches = [{"observation": torch.randn(B, cfg.obs_seq_len, cfg.obs_dim, device=DEV),
                "action": torch.randn(B, cfg.obs_seq_len, cfg.act_dim, device=DEV),
                "goal_observation": torch.randn(B, cfg.goal_seq_len, cfg.obs_dim, device=DEV)} for _ in range(4)]
    results = {}
    from autograd_reference import use_autograd_training
    for mode in ("1", "0"):
        if mode == "0":
            use_autograd_training(monkeypatch)
        agent = build_agent(cfg, lambda: make_module(cfg, w, "fp32"), device=DEV)
        agent.get_scaler(Scaler(np.random.default_rng(0).standard_normal((64, cfg.obs_dim)).astype(np.float32),
                                np.random.default_rng(1).standard_normal((64, cfg.act_dim)).astype(np.float32), True, DEV))
        agent.set_bounds(agent.scaler)
        monkeypatch.setattr(agent, "make_sample_density", lambda: (lambda shape, device: sigma.clone()))
        monkeypatch.setattr(torch, "randn_like", lambda t, *a, **k: noise.clone())
        losses = [agent.train_step(b) for b in batches]
        monkeypatch.undo()
        hip = getattr(agent, "_hip_step", None)
        assert (hip is not None) == (mode == "1")
        if hip is not None:
            flat = hip.flat_grads()
            assert flat is not None and flat.numel() == sum(p.numel() for p in agent.model.parameters())
        results[mode] = (losses, [p.detach().cpu().numpy().copy() for p in agent.model.parameters()])
    assert np.allclose(results["1"][0], results["0"][0], rtol=2e-5, atol=1e-6), (results["1"][0], results["0"][0])
    for a, b in zip(results["1"][1], results["0"][1]):
        assert rel_err(a, b) < 5e-3            # Adam amplifies last-bit differences to O(lr), see above


@pytest.mark.gpu
def test_training_with_the_hip_step_converges_like_autograd(monkeypatch):
    """300 train_steps on a fixed synthetic data set (kitchen-like dropouts on): the bf16 HIP step and the fp32
    torch-autograd step start from the same weights and must both bring the score-matching loss down, to the same
    level (the two use different dropout masks and the HIP one bf16 GEMM operands, so curves agree statistically)."""
    from test_host_logic import build_agent
    from beso_amd.networks.scaler.scaler_class import Scaler
    cfg = O.TINY
    w = O.make_weights(cfg, seed=5, std=0.02)
    g = torch.Generator(device="cpu").manual_seed(3)
    N, B = 512, 128
    obs = torch.randn(N, cfg.obs_seq_len, cfg.obs_dim, generator=g)
    goal = torch.randn(N, cfg.goal_seq_len, cfg.obs_dim, generator=g)
    # actions are a fixed smooth function of the observations and goals: something the network can fit
    proj = torch.randn(cfg.obs_dim, cfg.act_dim, generator=g) * 0.5
    act = torch.tanh(obs @ proj + (goal.mean(1, keepdim=True) @ proj) * 0.5)
    curves = {}
    from autograd_reference import use_autograd_training
    for mode, precision in (("1", "bf16"), ("0", "fp32")):
        if mode == "0":
            use_autograd_training(monkeypatch)
        torch.manual_seed(17)
        agent = build_agent(cfg, lambda: _train_module(cfg, w, precision, attn_pdrop=0.3, resid_pdrop=0.05), device=DEV, lr=2e-3)
        agent.get_scaler(Scaler(obs.reshape(-1, cfg.obs_dim).numpy(), act.reshape(-1, cfg.act_dim).numpy(), True, DEV))
        agent.set_bounds(agent.scaler)
        losses = []
        for it in range(300):
            idx = torch.randint(0, N, (B,), generator=g)
            losses.append(agent.train_step({"observation": obs[idx].to(DEV), "action": act[idx].to(DEV),
                                            "goal_observation": goal[idx].to(DEV)}))
        assert (getattr(agent, "_hip_step", None) is not None) == (mode == "1")
        curves[mode] = (float(np.mean(losses[:20])), float(np.mean(losses[-40:])))
    (h0, h1), (a0, a1) = curves["1"], curves["0"]
    assert h1 < 0.6 * h0 and a1 < 0.6 * a0, curves                 # both learn
    assert abs(h1 - a1) < 0.2 * a1, curves                          # ... to the same level


@pytest.mark.gpu
def test_training_step_random_shapes():
    """tools/fuzz_train.py for a few seconds: random model shapes (heads, head dims, windows, goal lengths incl. none,
    layer counts, ragged batches; short- and general-sequence attention kernels) against autograd, fp32 mode."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_train.py"), "8", "11"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "fuzz_train:" in r.stdout


@pytest.mark.gpu
def test_hip_loss_node_backward_twice():
    """The autograd node of the HIP training step keeps its gradients intact: two backward passes through a retained
    graph accumulate twice the gradient."""
    cfg = O.TINY
    m = _train_module(cfg, O.make_weights(cfg, seed=3, std=0.06), "fp32")
    state, action, goal, noise, sigma = _train_inputs(cfg, 6, seed=4)
    loss = m.loss(state, action, goal, noise, sigma)
    loss.backward(retain_graph=True)
    g1 = [p.grad.clone() for p in m.parameters()]
    loss.backward()
    for p, g in zip(m.parameters(), g1):
        assert torch.allclose(p.grad, 2 * g, rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_hip_train_steps_match_the_reference_agent_trace():
    """BesoAgent.train_step x 4 entirely in HIP (beso_loss_grad + beso_adam_step with the EMA warm-up rule, fp32 mode)
    against the trace that the REFERENCE's own agent produced on the same batches, noise and sigma
    (tests/golden/tiny_train_trace.npz): losses, final parameters and EMA shadow."""
    from test_host_logic import build_agent, replay_train_trace
    from beso_amd.optim import FusedAdam
    fx = load_golden("tiny_train_trace.npz")
    cfg = O.TINY
    w = O.make_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"]))
    agent = build_agent(cfg, lambda: make_module(cfg, w, "fp32"), device=DEV)
    agent.ema_helper.load_shadow_params(agent.model.get_params())
    assert isinstance(agent.optimizer, FusedAdam)
    loss_err, p_err, e_err = replay_train_trace(agent, fx, device=DEV)
    assert getattr(agent, "_hip_step", None) is not None
    assert max(loss_err) < 5e-5, loss_err
    assert p_err < 5e-3 and e_err < 5e-3, (p_err, e_err)      # Adam amplifies last-bit gradient differences to O(lr)


@pytest.mark.gpu
def test_hip_loss_pred_last_action_only(monkeypatch):
    """GCDenoiser.loss(..., pred_last_action_only=True) (score_wrappers.py:59-63,76-77: the noise of all but the last
    step is zeroed in place and only the last step is scored) through the HIP step against autograd."""
    cfg = O.TINY
    m = _train_module(cfg, O.make_weights(cfg, seed=3, std=0.06), "fp32")
    state, action, goal, noise, sigma = _train_inputs(cfg, 7, seed=6)
    out = {}
    from autograd_reference import loss_autograd
    for mode in ("0", "1"):
        for p in m.parameters():
            p.grad = None
        nz = noise.clone()
        if mode == "0":
            loss = loss_autograd(m, state, action, goal, nz, sigma, pred_last_action_only=True)
        else:
            loss = m.loss(state, action, goal, nz, sigma, pred_last_action_only=True)
        assert ("ScoreMatchingLoss" in type(loss.grad_fn).__name__) == (mode == "1")
        assert float(nz[:, :-1].abs().max()) == 0.0 and float(nz[:, -1].abs().max()) > 0.0       # mutated like the reference
        loss.backward()
        out[mode] = (loss.item(), [p.grad.clone() for p in m.parameters()])
    assert abs(out["1"][0] - out["0"][0]) < 2e-5 * abs(out["0"][0])
    assert max(_grad_errors(out["1"][1], out["0"][1])) < 2e-4


# ------------------------------------------------------------------------------------------------
# training feed (beso_gather_windows)
# ------------------------------------------------------------------------------------------------
def _feed(fx, mode, batch_size=16, **kw):
    from beso_amd.data.trajectory_feed import DeviceTrajectoryFeed
    sub = fx["subset"]
    cond = mode != "none"
    return DeviceTrajectoryFeed(fx["observations"][sub], fx["actions"][sub], fx["lengths"][sub], int(fx["window"]), batch_size, DEV,
                                future_conditional=cond, min_future_sep=int(fx["min_future_sep"]),
                                future_seq_len=int(fx["future_seq_len"]) if cond else None,
                                only_sample_tail=mode == "tail", only_sample_seq_end=mode == "seq_end", **kw)


@pytest.mark.parametrize("mode", ["none", "random", "tail", "seq_end"])
def test_feed_gather_vs_reference_items(mode):
    """beso_gather_windows against items of the reference's TrajectorySlicerDataset (collated), bit for bit; the random
    future-goal starts are the ones np.random gave the reference."""
    fx = load_golden("trajectory_windows.npz")
    feed = _feed(fx, mode)
    np.testing.assert_array_equal(feed.slices, fx["slices"])
    batch = feed.gather(torch.from_numpy(fx["ids"]), torch.from_numpy(fx["random::draws"]) if mode == "random" else None)
    assert set(batch) == {"observation", "action"} | ({"goal_observation"} if mode != "none" else set())
    for key, val in batch.items():
        assert val.is_cuda and val.dtype == torch.float32
        np.testing.assert_array_equal(val.cpu().numpy(), fx[f"{mode}::{key}"], err_msg=f"{mode}::{key}")


def test_feed_epoch_and_oracle_at_size():
    """A kitchen-sized dataset: every window exactly once per epoch (also split over two ranks), batches equal the
    oracle's slicing of the same ids / draws, a new epoch is a new permutation, out-of-range ids give zero rows."""
    from beso_amd.data.trajectory_feed import DeviceTrajectoryFeed
    rng = np.random.default_rng(0)
    n, t_max, obs, act, window, glen = 120, 200, 30, 9, 10, 1
    lengths = rng.integers(5, t_max + 1, size=n).astype(np.int32)
    observations = rng.standard_normal((n, t_max, obs)).astype(np.float32)
    actions = rng.standard_normal((n, t_max, act)).astype(np.float32)
    table = O.window_table(lengths, window)
    kw = dict(future_conditional=True, future_seq_len=glen, min_future_sep=3, seed=5)
    feed = DeviceTrajectoryFeed(observations, actions, lengths, window, 1024, DEV, **kw)
    assert feed.n_windows == len(table) and len(feed) == -(-len(table) // 1024)
    ids = torch.from_numpy(rng.integers(0, len(table), size=4096))
    draws = torch.from_numpy(rng.integers(0, 2 ** 62, size=4096))
    got = feed.gather(ids, draws)
    want = O.slice_windows(observations, actions, lengths, table, ids.numpy(), glen, 3, "random", draws.numpy())
    for key in want:
        np.testing.assert_array_equal(got[key].cpu().numpy(), want[key], err_msg=key)
    # an epoch covers every window once: the observation windows, as a multiset, are the table's
    def epoch_keys(f):
        rows = torch.cat([b["observation"][:, 0, :2] for b in f])          # (first row's first two features) identify a window
        return rows.cpu().numpy()
    all_first = np.stack([observations[i, s, :2] for i, s, _ in table])
    e1, e2 = epoch_keys(feed), epoch_keys(feed)
    assert e1.shape == all_first.shape and not np.array_equal(e1, e2)
    order = lambda a: a[np.lexsort(a.T)]                                   # noqa: E731
    np.testing.assert_array_equal(order(e1), order(all_first))
    np.testing.assert_array_equal(order(e2), order(all_first))
    # data-parallel shares: the same number of windows AND of batches on every rank (a rank with one batch more would
    # all-reduce alone), every window covered, the surplus (world * ceil(n / world) - n) wrapped from the front
    for world in (2, 3, 8):
        feeds = [DeviceTrajectoryFeed(observations, actions, lengths, window, 256, DEV, rank=r, world_size=world, **kw)
                 for r in range(world)]
        assert len({len(f) for f in feeds}) == 1
        parts = [epoch_keys(f) for f in feeds]
        assert len({len(p) for p in parts}) == 1 and len(parts[0]) == -(-len(table) // world)
        both = np.concatenate(parts)
        uniq = np.unique(both, axis=0)
        np.testing.assert_array_equal(order(uniq), order(np.unique(all_first, axis=0)))
        assert len(both) - len(all_first) == world * len(parts[0]) - len(table) < world
    bad = feed.gather(torch.tensor([-1, len(table), 3]))
    assert float(bad["observation"][:2].abs().max()) == 0.0 and float(bad["observation"][2].abs().max()) > 0.0
    # drop_last and no shuffling
    seq = DeviceTrajectoryFeed(observations, actions, lengths, window, 1000, DEV, shuffle=False, drop_last=True)
    first = next(iter(seq))
    assert len(seq) == len(table) // 1000 and first["observation"].shape == (1000, window, obs) and "goal_observation" not in first
    np.testing.assert_array_equal(first["action"].cpu().numpy(), np.stack([actions[i, s:e] for i, s, e in table[:1000]]))


def test_feed_from_sliced_and_train_step():
    """from_sliced on an object with the reference slicer's attributes, and the dict batches drive train_step."""
    from beso_amd.data.trajectory_feed import DeviceTrajectoryFeed
    from test_host_logic import build_agent
    fx = load_golden("trajectory_windows.npz")
    cfg = O.TINY

    class Base:
        def __init__(self, obs, act, lengths):
            self.obs, self.act, self.lengths = obs, act, lengths

        def __len__(self):
            return len(self.lengths)

        def __getitem__(self, i):
            return torch.from_numpy(self.obs[i]), torch.from_numpy(self.act[i]), torch.ones(self.obs.shape[1])

        def get_seq_length(self, i):
            return int(self.lengths[i])

    rng = np.random.default_rng(1)
    n, t_max = 12, 40
    lengths = rng.integers(cfg.obs_seq_len, t_max + 1, size=n)
    base = Base(rng.standard_normal((n, t_max, cfg.obs_dim)).astype(np.float32),
                rng.standard_normal((n, t_max, cfg.act_dim)).astype(np.float32), lengths)
    table = O.window_table(lengths, cfg.obs_seq_len)

    class Sliced:
        dataset, window, future_conditional, min_future_sep = base, cfg.obs_seq_len, True, 0
        future_seq_len, only_sample_tail, only_sample_seq_end = cfg.goal_seq_len, False, False
        slices = [tuple(r) for r in table]

    feed = DeviceTrajectoryFeed.from_sliced(Sliced, 64, DEV, seed=1)
    assert feed.n_windows == len(table)
    agent = build_agent(cfg, lambda: make_module(cfg, O.make_weights(cfg, seed=1, std=0.05), "fp32"), device=DEV)
    from beso_amd.networks.scaler.scaler_class import Scaler
    agent.get_scaler(Scaler(base.obs.reshape(-1, cfg.obs_dim), base.act.reshape(-1, cfg.act_dim), True, DEV))
    agent.set_bounds(agent.scaler)
    losses = [agent.train_step(b) for b in feed]
    assert len(losses) == len(feed) and all(np.isfinite(v) for v in losses)
    # the reference's step loop, fed and evaluated from HBM, across an epoch boundary
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        agent.working_dir, agent.max_train_steps, agent.eval_every_n_steps = tmp, len(feed) + 2, len(feed)
        test_feed = DeviceTrajectoryFeed.from_sliced(Sliced, 128, DEV, shuffle=False)
        before = agent.steps
        agent.train_agent_on_steps(feed, test_feed)
        assert agent.steps == before + len(feed) + 2 and os.path.exists(os.path.join(tmp, "model_state_dict.pth"))
    Sliced.slices = Sliced.slices[:-1]
    with pytest.raises(ValueError):
        DeviceTrajectoryFeed.from_sliced(Sliced, 64, DEV)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


# ------------------------------------------------------------------------------------------------
# overlapped gradient exchange (beso_loss_grad_overlap)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg_name,precision", [("kitchen", "bf16"), ("tiny", "fp32"), ("tiny_mlp_head", "fp32")])
def test_early_gradient_range_is_final_when_the_early_stream_runs(cfg_name, precision):
    """The gradients with the early range released to a second stream equal the plain call's, and a copy of the early
    range enqueued on that stream right after the call already holds the final values (the stream was ordered behind
    their completion, not behind the whole backward)."""
    from beso_amd.training import HipTrainStep
    cfg = O.CONFIGS[cfg_name]
    m = _train_module(cfg, O.make_weights(cfg, seed=3, std=0.05), precision)
    B = 256 if cfg_name == "kitchen" else 9
    state, action, goal, noise, sigma = _train_inputs(cfg, B, seed=4)
    step = HipTrainStep(m.inner_model, cfg.sigma_data)
    b, e = step.early_range()
    if cfg.n_layers < 2:
        assert b == e                                                   # nothing is early; the call must still work
    else:
        assert 0 < b < e < step.n_grad
    loss0, flat0, _ = step.run(state, action, goal, noise, sigma, seed=11, fresh_grads=True)
    side = torch.cuda.Stream(DEV)
    early = torch.empty(e - b, device=DEV)
    for _ in range(3):                                                  # (repeat: the event object is reused across calls)
        loss1, flat1, _ = step.run(state, action, goal, noise, sigma, seed=11, fresh_grads=True, early_stream=side)
        with torch.cuda.stream(side):
            early.copy_(flat1[b:e], non_blocking=True)
        torch.cuda.synchronize()
        assert torch.equal(early, flat1[b:e]), "the early range changed after the early stream was released"
        assert e == b or float(early.abs().max()) > 0.0
        scale = float(flat0.abs().max())
        assert float((flat1 - flat0).abs().max()) <= 2e-6 * scale      # (bias sums accumulate with atomics)
        assert abs(float(loss1) - float(loss0)) <= 1e-6 * abs(float(loss0))


def test_overlapped_all_reduce_on_a_one_rank_group():
    """all_reduce_sum_overlapped through RCCL (a one-rank group on this GPU: the collectives are identities, the stream
    and work-handle choreography is the real one), driven by BesoAgent.train_step's data-parallel branch."""
    import torch.distributed as dist
    from beso_amd import distributed as bdist
    from test_host_logic import build_agent
    from beso_amd.networks.scaler.scaler_class import Scaler
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    real = bdist.is_distributed
    try:
        cfg = O.TINY
        w = O.make_weights(cfg, seed=1, std=0.05)
        torch.manual_seed(3)
        batch = {"observation": torch.randn(32, cfg.obs_seq_len, cfg.obs_dim, device=DEV),
                 "action": torch.randn(32, cfg.obs_seq_len, cfg.act_dim, device=DEV),
                 "goal_observation": torch.randn(32, cfg.goal_seq_len, cfg.obs_dim, device=DEV)}
        params = {}
        for overlap in ("1", "0", "sharded"):
            os.environ["BESO_AMD_C1_OVERLAP"] = "1" if overlap == "sharded" else overlap
            os.environ["BESO_AMD_C1"] = "sharded" if overlap == "sharded" else "overlap"
            bdist.is_distributed = lambda: True                         # one rank, but take the data-parallel branch
            real_sum = bdist.all_reduce_sum_overlapped
            calls = []
            bdist.all_reduce_sum_overlapped = lambda f, r, st: (calls.append(r), real_sum(f, r, st, _single_rank_ok=True))
            agent = build_agent(cfg, lambda: make_module(cfg, w, "fp32"), device=DEV)
            agent.get_scaler(Scaler(np.random.default_rng(0).standard_normal((64, cfg.obs_dim)).astype(np.float32),
                                    np.random.default_rng(1).standard_normal((64, cfg.act_dim)).astype(np.float32), True, DEV))
            agent.set_bounds(agent.scaler)
            torch.manual_seed(9)
            losses = [agent.train_step(batch) for _ in range(3)]
            bdist.all_reduce_sum_overlapped = real_sum
            assert (len(calls) == 3) == (overlap == "1") and all(np.isfinite(v) for v in losses)
            if overlap == "sharded":
                # reduce-scatter -> Adam(W) + EMA on the owned range (here: everything) -> all-gather, through RCCL
                assert agent._sharded_ex is not None and (agent._sharded_ex.lo, agent._sharded_ex.hi) == (0, agent._sharded_ex.n)
                assert agent._ema_partial
                agent.evaluate(batch)                                   # reads the EMA: completes it first (a collective)
                assert not agent._ema_partial
            params[overlap] = (torch.cat([q.detach().reshape(-1) for q in agent.model.get_params()]), losses)
        # same trajectory: equal losses; equal parameters except where the true gradient is zero (key biases: Adam turns
        # the atomics' rounding noise into +-lr there)
        assert np.allclose(params["1"][1], params["0"][1], rtol=1e-4), (params["1"][1], params["0"][1])
        diff = (params["1"][0] - params["0"][0]).abs()
        print(f"[overlap] losses {params['1'][1]} vs {params['0'][1]}; parameters differing by > 1e-6: {float((diff > 1e-6).float().mean()):.4f}")
        assert float((diff > 1e-6).float().mean()) < 0.03 and float(diff.max()) < 1e-3
        assert np.allclose(params["sharded"][1], params["0"][1], rtol=1e-4)
        diff = (params["sharded"][0] - params["0"][0]).abs()
        assert float((diff > 1e-6).float().mean()) < 0.03 and float(diff.max()) < 1e-3
    finally:
        bdist.is_distributed = real
        os.environ.pop("BESO_AMD_C1_OVERLAP", None)
        os.environ.pop("BESO_AMD_C1", None)
        dist.destroy_process_group()


def test_fused_euler_ancestral_loop():
    """beso_sample_ancestral (euler_ancestral as one enqueue): (i) with the reference run's recorded noise it reproduces the
    reference output; (ii) with a seeded generator it equals the step-by-step loop (same randn_like sequence), for the plain
    denoiser and for a classifier-free pair, through the fused bf16 kernels too."""
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    fx = load_golden("tiny_euler_ancestral.npz")
    cfg = O.TINY
    m = make_module(cfg, O.make_weights(cfg, seed=int(fx["seed"]), std=0.02), "fp32")
    with torch.no_grad():
        out = m.fused_sampler("euler_ancestral", G(fx["state"]), G(fx["x_t"]), G(fx["goal"]), fx["sigmas"], noise=G(fx["noise"]))
        assert out is not None and rel_err(out.cpu().numpy(), fx["out"]) < 2e-5
        for cfg_name, precision, wrap in (("tiny", "fp32", None), ("kitchen", "bf16", None), ("block_push", "fp32", 2.0)):
            c = O.CONFIGS[cfg_name]
            mm = make_module(c, O.make_weights(c, seed=2, std=0.04), precision)
            model = mm if wrap is None else ClassifierFreeSampleModel(mm, wrap)
            s, g, x = (G(v) for v in O.make_inputs(c, 6, seed=9))
            sig = ks.get_sigmas_exponential(5, 0.05, 1.0)
            keep = x.clone()
            torch.manual_seed(77)
            fused = ks.sample_euler_ancestral(model, s, x, g, sig, disable=True)
            assert torch.equal(x, keep)
            torch.manual_seed(77)
            loop = ks.sample_euler_ancestral(model, s, x, g, sig, disable=True, callback=lambda info: None)     # generic loop
            err = rel_err(fused.cpu().numpy(), loop.cpu().numpy())
            print(f"[parity] fused euler_ancestral vs loop {cfg_name} {precision}: {err:.3e}")
            assert err < (2e-6 if precision == "fp32" else 2e-2)


def test_feed_takes_rank_and_seed_from_the_process_group():
    """Under torchrun the feed reads rank / world size from the process group and shares rank 0's shuffling seed
    (a one-rank RCCL group here: the broadcast really runs)."""
    import torch.distributed as dist
    from beso_amd import distributed as bdist
    from beso_amd.data.trajectory_feed import DeviceTrajectoryFeed
    fx = load_golden("trajectory_windows.npz")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    real = bdist.is_distributed
    try:
        bdist.is_distributed = lambda: True
        sub = fx["subset"]
        feed = DeviceTrajectoryFeed(fx["observations"][sub], fx["actions"][sub], fx["lengths"][sub], int(fx["window"]), 16, DEV)
        assert (feed.rank, feed.world_size) == (0, 1) and len(feed) == -(-feed.n_windows // 16)
        assert sum(b["observation"].shape[0] for b in feed) == feed.n_windows
    finally:
        bdist.is_distributed = real
        dist.destroy_process_group()


# -------------------------------------------------------------------------------------------------
# round 5: the chip-wide small-batch path (small.hip) -- what the library runs BY ITSELF up to 448 token rows in bf16 and 4096 in fp32
# -------------------------------------------------------------------------------------------------
_SMALL_FORWARD = [("tiny_forward.npz", "tiny"), ("tiny_mlp_head_forward.npz", "tiny_mlp_head"), ("tiny_nogoal_forward.npz", "tiny_nogoal"),
                  ("kitchen_forward_std002.npz", "kitchen"), ("kitchen_forward_std008.npz", "kitchen"), ("block_push_forward.npz", "block_push")]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("fixture,cfg_name", _SMALL_FORWARD)
def test_small_batch_path_vs_reference_vectors(fixture, cfg_name, precision):
    """The library's OWN choice at the fixtures' batch sizes (no plan hint: the rollout workload of the reference is B = 1) is
    the chip-wide small-batch path -- four short launches per layer (three up to 96 token rows in bf16: the out-projection rides
    in the attention launch) that spread every weight matrix over the CUs (small.hip).
    GCDenoiser.forward / DiffusionGPT.forward, t in {1, W/2, W}, cond and uncond against the reference's vectors: fp32
    (exact-fp32 MFMA, two-pass LayerNorm, erff) 2e-5, bf16 inside the bounds of the one-launch kernel on the same vectors;
    every call is one timed region at the small-batch site and none at the fused kernel's."""
    from beso_amd import _lib
    from beso_amd.runtime import set_plan
    # the library's own choice: fp32 -- every shape here; bf16 -- where there are weights enough to spread (kitchen: 6 layers of
    # 360^2; block-push and the test-sized models stay on the one-launch / per-op kernels, and take the path by hint here)
    own = precision == "fp32" or cfg_name == "kitchen"
    set_plan(forward=0 if own else _lib.PLAN_SMALL)
    fx = load_golden(fixture)
    cfg = O.CONFIGS[cfg_name]
    m = make_module(cfg, _weights(fx, cfg), precision)
    worst, calls = [0.0], [0]

    def run():
        for t in fx["ts"]:
            p = f"t{int(t)}::"
            s, a, g, sg = (G(fx[p + k]) for k in ("state", "action", "goal", "sigma"))
            e1 = rel_err(m(s, a, g, sg).cpu().numpy(), fx[p + "denoised"])
            e2 = rel_err(m(s, a, g, sg, uncond=True).cpu().numpy(), fx[p + "denoised_uncond"])
            e3 = rel_err(m.inner_model(s, a, g, sg).cpu().numpy(), fx[p + "inner"])
            worst[0] = max(worst[0], e1, e2, e3)
            calls[0] += 3

    with torch.no_grad():
        n_small = count_site_launches("small", run)
        calls[0] = 0
        n_fused = count_fused_launches(run)
    print(f"[parity] small-batch path {fixture} {precision}: max rel err {worst[0]:.3e} ({calls[0]} calls)")
    assert n_small == calls[0] and n_fused == 0, (n_small, n_fused, calls[0])
    # bf16: this path rounds where the per-op bf16 kernels round (LayerNorm output, q | k | v, attention output, GELU(h) as bf16
    # tensors; the one-launch kernel keeps more of them in fp32 registers): its own table, twice what it measures
    assert worst[0] < (TOL_BF16_SMALL.get(fixture, TOL[precision]) if precision == "bf16" else TOL[precision])


TOL_BF16_SMALL = {"kitchen_forward_std002.npz": 1.5e-2, "kitchen_forward_std008.npz": 2e-2, "block_push_forward.npz": 4.2e-2,
                  "tiny_forward.npz": 8e-3, "tiny_mlp_head_forward.npz": 1.1e-2, "tiny_nogoal_forward.npz": 1.7e-2}      # measured: 7.2e-3, 9.8e-3, 2.1e-2, 3.6e-3, 5.1e-3, 8.2e-3


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_small_batch_path_samplers_cfg_and_hints(precision):
    """Sampler loops and classifier-free guidance at the fixtures' batch sizes through the library's own choice (every
    evaluation on the small-batch path: none at the fused kernel's site) against the reference's sampler outputs; and the plan
    hints around it: BESO_PLAN_SMALL takes the path at a batch the library would give to the one-launch kernel, BESO_PLAN_FUSED
    / BESO_PLAN_PER_OP leave it -- same results inside the precision's bound either way."""
    from beso_amd import _lib
    from beso_amd.runtime import set_plan
    from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
    from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
    fns = {"ddim": ks.sample_ddim, "euler": ks.sample_euler, "heun": ks.sample_heun, "dpmpp_2m": ks.sample_dpmpp_2m,
           "dpm": ks.sample_dpm_2, "dpmpp_2s": ks.sample_dpmpp_2s}
    for fixture, cfg_name in [("kitchen_samplers.npz", "kitchen"), ("block_push_heun_cfg.npz", "block_push")]:
        set_plan(forward=0 if (precision == "fp32" or cfg_name == "kitchen") else _lib.PLAN_SMALL)      # (bf16 block-push: by hint)
        fx = load_golden(fixture)
        cfg = O.CONFIGS[cfg_name]
        m = make_module(cfg, _weights(fx, cfg), precision)
        lam = float(fx["cond_lambda"])
        model = m if lam < 0 else ClassifierFreeSampleModel(m, lam)
        for key in sorted(k[:-5] for k in fx if k.endswith("::out")):
            n = int(key.split("_")[-2])
            sampler = key[: key.index(f"_{n}_")]
            out = [None]
            nf = count_fused_launches(lambda: out.__setitem__(0, fns[sampler](
                model, G(fx["state"]), G(fx["x_t"]), G(fx["goal"]), torch.from_numpy(fx[key + "::sigmas"]), disable=True)))
            err = rel_err(out[0].cpu().numpy(), fx[key + "::out"])
            print(f"[parity] small-batch path {fixture}:{key} {precision}: {err:.3e}")
            assert nf == 0, (key, nf)
            assert err < (TOL[precision] * (1 if n <= 10 else 8) if precision == "fp32" else 2e-2), key
    fx = load_golden("block_push_cfg.npz")
    m = make_module(O.BLOCK_PUSH, _weights(fx, O.BLOCK_PUSH), precision)
    s, a, g, sg = (G(fx[k]) for k in ("state", "action", "goal", "sigma"))
    set_plan(forward=_lib.PLAN_SMALL)
    with torch.no_grad():
        for lam in fx["lambdas"]:
            box = [None]
            n = count_site_launches("small", lambda: box.__setitem__(0, ClassifierFreeSampleModel(m, float(lam))(s, a, g, sg)))
            assert n == 1 and rel_err(box[0].cpu().numpy(), fx[f"lam{float(lam)}"]) < TOL[precision], lam
        # hints: 400 kitchen samples = 4400 token rows -- beyond the library's own thresholds
        cfg = O.KITCHEN
        mk = make_module(cfg, O.make_weights(cfg, seed=4, std=0.04), precision)
        s, g, a = (G(v) for v in O.make_inputs(cfg, 400, seed=3))
        sg = G(np.linspace(0.05, 0.9, 400).astype(np.float32))
        outs = {}
        for name, hint in (("own", 0), ("small", _lib.PLAN_SMALL), ("other", _lib.PLAN_FUSED if precision == "bf16" else _lib.PLAN_PER_OP)):
            set_plan(forward=hint)
            n = count_site_launches("small", lambda: outs.__setitem__(name, mk(s, a, g, sg)))
            assert n == (1 if name == "small" else 0), (name, n)
        set_plan(forward=0)
        assert torch.equal(outs["own"], outs["other"])
        dev_rel = rel_err(outs["small"].cpu().numpy(), outs["other"].cpu().numpy())
        print(f"[parity] small-batch path vs the library's kernels at B = 400, {precision}: {dev_rel:.2e}")
        assert dev_rel < (2e-5 if precision == "fp32" else 2e-2)
        # ragged: B = 1, 3, 33 (token rows not a multiple of the 32-row tile), short windows; B = 64 (BASELINE configs[0]) and 200:
        # the wide bf16 instances of round 6 (32 rows x 128 / 32 features, a head in twelve waves)
        for B, t in ((1, 1), (3, 2), (33, cfg.obs_seq_len), (64, cfg.obs_seq_len), (200, cfg.obs_seq_len), (93, 2)):
            s, g, a = (G(v) for v in O.make_inputs(cfg, B, seed=B, t=t))
            sg = G(np.linspace(0.1, 0.8, B).astype(np.float32))
            set_plan(forward=_lib.PLAN_SMALL)
            o_small = mk(s, a, g, sg)
            set_plan(forward=_lib.PLAN_PER_OP)
            o_ref = mk(s, a, g, sg)
            set_plan(forward=_lib.PLAN_SMALL)
            assert rel_err(o_small.cpu().numpy(), o_ref.cpu().numpy()) < (2e-5 if precision == "fp32" else 2e-2), (B, t)
            assert torch.equal(o_small, mk(s, a, g, sg)), "the small-batch path is deterministic"
        set_plan(forward=0)


def test_scaler_scale_many_is_one_launch_with_the_reference_bits():
    """Scaler.scale_many (beso_scale_rows): state, goal and action of a training batch in one launch -- bit for bit the
    (x - mean) / (std + 1e-12) of scale_input / scale_output (scaler_class.py:95-117), ragged sizes included; CPU tensors and
    float64 statistics take the per-tensor methods."""
    from beso_amd.networks.scaler.scaler_class import Scaler
    rng = np.random.default_rng(3)
    sc = Scaler(rng.standard_normal((64, 30)).astype(np.float32) * 3 + 1, rng.standard_normal((64, 9)).astype(np.float32), True, DEV)
    for B in (1, 7, 1024):
        st, go, ac = (torch.randn(B, 5, 30, device=DEV) * 2, torch.randn(B, 2, 30, device=DEV), torch.randn(B, 5, 9, device=DEV))
        got = sc.scale_many([(st, "x"), (go, "x"), (ac, "y")])
        ref = [sc.scale_input(st), sc.scale_input(go), sc.scale_output(ac)]
        assert all(torch.equal(a, b) for a, b in zip(got, ref)), B
    cpu = sc.scale_many([(st.cpu(), "x")])[0]
    assert torch.equal(cpu.to(DEV), sc.scale_input(st))
    sc64 = Scaler(rng.standard_normal((64, 30)), rng.standard_normal((64, 9)), True, DEV)      # float64 statistics: the fallback
    assert sc64.scale_many([(st, "x")])[0].dtype == torch.float32


def test_small_batch_path_fuzz_for_ten_seconds():
    """tools/fuzz_small.py for ten seconds: random model shapes, batches (1 ... 300 samples: the 16-row, 32-row and wide instances),
    windows and guidance through the small-batch path against the per-op kernels of the same precision."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_small", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_small.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, worst = mod.run(10.0, 6)
    print(f"[parity] fuzz_small: {n} cases, fp32 {worst['fp32']:.2e}, bf16 {worst['bf16']:.2e}")
    assert n >= 20


def test_small_batch_path_serves_the_agent_rollout():
    """BesoAgent.predict at B = 1 (kitchen_workspace_manager.py:286-294) with no plan hint: the reference's six-call trace
    through the EMA packed image, every evaluation on the small-batch path."""
    from beso_amd.runtime import set_plan
    from test_host_logic import build_agent, run_agent_trace
    set_plan(forward=0)
    fx = load_golden("tiny_agent_trace.npz")
    cfg = O.TINY
    w = weights_from_fixture(fx)
    agent = build_agent(cfg, lambda: make_module(cfg, w, "fp32"), device=DEV)
    agent.ema_helper.load_shadow_params(agent.model.get_params())
    err = [None]
    n = count_site_launches("small", lambda: err.__setitem__(0, run_agent_trace(agent, fx, device=DEV)))
    assert err[0] < 5e-5 and n > 0, (err[0], n)
