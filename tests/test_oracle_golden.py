"""Pin the CPU oracle (oracle/beso_oracle.py) against vectors produced by the reference itself
(tests/golden/make_fixtures.py).  The reference has no tests of its own for this path."""
import numpy as np
import pytest

from oracle import beso_oracle as O
from conftest import load_golden, weights_from_fixture, rel_err

TOL = 2e-5   # fp32 oracle (numpy/OpenBLAS) vs fp32 reference (torch CPU): summation-order noise only


def _weights(fx, cfg):
    w = weights_from_fixture(fx)
    if not w:
        w = O.make_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"]))
        ws = sum(float(np.abs(v.astype(np.float64)).sum()) for v in w.values())
        assert abs(ws - float(fx["wsum"])) <= 1e-9 * ws, "seeded weight recipe drifted"
    return w


@pytest.mark.parametrize("fixture,cfg_name", [
    ("tiny_forward.npz", "tiny"), ("tiny_mlp_head_forward.npz", "tiny_mlp_head"),
    ("tiny_nogoal_forward.npz", "tiny_nogoal"), ("kitchen_forward_std002.npz", "kitchen"),
    ("kitchen_forward_std008.npz", "kitchen"), ("block_push_forward.npz", "block_push"),
    ("long_horizon_forward.npz", "long_horizon")])
def test_forward_matches_reference(fixture, cfg_name):
    fx = load_golden(fixture)
    cfg = O.CONFIGS[cfg_name]
    w = _weights(fx, cfg)
    for t in fx["ts"]:
        p = f"t{int(t)}::"
        s, g, a, sig = fx[p + "state"], fx[p + "goal"], fx[p + "action"], fx[p + "sigma"]
        assert rel_err(O.score_gpt_forward(w, cfg, s, a, g, sig), fx[p + "inner"]) < TOL
        assert rel_err(O.denoise(w, cfg, s, a, g, sig), fx[p + "denoised"]) < TOL
        assert rel_err(O.denoise(w, cfg, s, a, g, sig, uncond=True), fx[p + "denoised_uncond"]) < TOL
        # fp64 arbiter agrees with the fp32 reference to fp32 round-off as well
        assert rel_err(O.denoise(w, cfg, s, a, g, sig, dtype=np.float64), fx[p + "denoised"]) < TOL


@pytest.mark.parametrize("fixture,cfg_name", [
    ("tiny_forward.npz", "tiny"), ("tiny_mlp_head_forward.npz", "tiny_mlp_head"),
    ("tiny_nogoal_forward.npz", "tiny_nogoal"), ("kitchen_forward_std002.npz", "kitchen"),
    ("block_push_forward.npz", "block_push"), ("long_horizon_forward.npz", "long_horizon")])
def test_aten_backend_of_the_oracle_matches_reference(fixture, cfg_name):
    """oracle/beso_oracle_torch.py (the ATen restatement that bench.py times as the CPU baseline) against the same
    reference vectors, forward and a DDIM loop."""
    import torch
    from oracle import beso_oracle_torch as OT
    fx = load_golden(fixture)
    cfg = O.CONFIGS[cfg_name]
    W = OT.to_torch(_weights(fx, cfg))
    T = lambda v: torch.from_numpy(np.ascontiguousarray(v))      # noqa: E731
    for t in fx["ts"]:
        p = f"t{int(t)}::"
        s, g, a, sig = (T(fx[p + k]) for k in ("state", "goal", "action", "sigma"))
        assert rel_err(OT.score_gpt_forward(W, cfg, s, a, g, sig).numpy(), fx[p + "inner"]) < TOL
        assert rel_err(OT.denoise(W, cfg, s, a, g, sig).numpy(), fx[p + "denoised"]) < TOL
        assert rel_err(OT.denoise(W, cfg, s, a, g, sig, uncond=True).numpy(), fx[p + "denoised_uncond"]) < TOL


def test_aten_backend_ddim_matches_reference():
    import torch
    from oracle import beso_oracle_torch as OT
    fx = load_golden("kitchen_samplers.npz")
    cfg = O.KITCHEN
    W = OT.to_torch(_weights(fx, cfg))
    T = lambda v: torch.from_numpy(np.ascontiguousarray(v))      # noqa: E731
    for key in ("ddim_3_exponential", "ddim_10_exponential"):
        out = OT.sample_ddim(W, cfg, T(fx["state"]), T(fx["x_t"]), T(fx["goal"]), T(fx[key + "::sigmas"]))
        assert rel_err(out.numpy(), fx[key + "::out"]) < TOL, key


def test_param_count_matches_survey():
    assert O.n_params(O.KITCHEN) == 9_381_249
    assert O.n_params(O.BLOCK_PUSH) == 2_783_762
    assert O.n_params(O.LONG_HORIZON) == 18_959_881
    assert O.KITCHEN.flops_per_sample() == 206_514_000
    assert O.BLOCK_PUSH.flops_per_sample() == 66_947_040
    assert O.LONG_HORIZON.flops_per_sample() == 2_585_961_472


@pytest.mark.parametrize("fixture,cfg_name", [
    ("kitchen_samplers.npz", "kitchen"), ("block_push_heun_cfg.npz", "block_push"),
    ("long_horizon_euler.npz", "long_horizon"), ("long_horizon_euler100.npz", "long_horizon")])
def test_samplers_match_reference(fixture, cfg_name):
    fx = load_golden(fixture)
    cfg = O.CONFIGS[cfg_name]
    w = _weights(fx, cfg)
    lam = float(fx["cond_lambda"])
    model = O.make_model(w, cfg, cond_lambda=None if lam < 0 else lam)
    keys = sorted(k[:-len("::out")] for k in fx if k.endswith("::out"))
    assert keys
    for key in keys:
        sampler, n, schedule = key.split("_")[0], int(key.split("_")[-2]), key.split("_")[-1]
        sampler = key[: key.index(f"_{n}_")]
        sig = fx[key + "::sigmas"]
        # schedules are restated too
        if schedule == "karras":
            mine = O.get_sigmas_karras(n, float(fx["sigma_min"]), float(fx["sigma_max"]), 5.0)
        else:
            mine = O.SCHEDULES[schedule](n, float(fx["sigma_min"]), float(fx["sigma_max"]))
        np.testing.assert_allclose(mine, sig, rtol=2e-6, atol=0)
        out = O.SAMPLERS[sampler](model, fx["state"], fx["x_t"], fx["goal"], sig)
        # many-step samplers accumulate fp32 round-off; 50-step Heun x CFG stays below 2e-4
        tol = 5e-5 if n <= 10 else 3e-4
        assert rel_err(out, fx[key + "::out"]) < tol, key


def test_euler_ancestral_with_injected_noise():
    fx = load_golden("tiny_euler_ancestral.npz")
    cfg = O.TINY
    w = O.make_weights(cfg, seed=int(fx["seed"]), std=0.02)
    out = O.sample_euler_ancestral(O.make_model(w, cfg), fx["state"], fx["x_t"], fx["goal"], fx["sigmas"],
                                   noise_list=fx["noise"])
    assert rel_err(out, fx["out"]) < TOL


def test_classifier_free_guidance():
    fx = load_golden("block_push_cfg.npz")
    cfg = O.BLOCK_PUSH
    w = _weights(fx, cfg)
    for lam in fx["lambdas"]:
        out = O.denoise_cfg(w, cfg, fx["state"], fx["action"], fx["goal"], fx["sigma"], float(lam))
        assert rel_err(out, fx[f"lam{float(lam)}"]) < TOL


@pytest.mark.parametrize("fixture,cfg_name", [("kitchen_loss.npz", "kitchen"), ("block_push_loss.npz", "block_push"),
                                              ("tiny_mlp_head_loss.npz", "tiny_mlp_head")])
def test_loss_at_the_shipped_shapes(fixture, cfg_name):
    fx = load_golden(fixture)
    cfg = O.CONFIGS[cfg_name]
    w = _weights(fx, cfg)
    loss = O.score_matching_loss(w, cfg, fx["state"], fx["action"], fx["goal"], fx["noise"], fx["sigma"])
    assert abs(float(loss) - float(fx["loss"])) < 1e-5 * abs(float(fx["loss"]))


def test_loss_and_sigma_density():
    fx = load_golden("tiny_loss.npz")
    cfg = O.TINY
    w = _weights(fx, cfg)
    loss = O.score_matching_loss(w, cfg, fx["state"], fx["action"], fx["goal"], fx["noise"], fx["sigma"])
    assert abs(float(loss) - float(fx["loss"])) < 1e-5 * abs(float(fx["loss"]))
    s = O.log_logistic_from_uniform(fx["loglogistic::u"], np.log(0.5), 0.5, 0.005, 1.0)
    np.testing.assert_allclose(s, fx["loglogistic::sigma"], rtol=1e-6)


def test_schedules():
    fx = load_golden("schedules.npz")
    fns = {"exponential": lambda n: O.get_sigmas_exponential(n, 0.005, 1.0),
           "linear": lambda n: O.get_sigmas_linear(n, 0.005, 1.0),
           "karras": lambda n: O.get_sigmas_karras(n, 0.005, 1.0, 5.0),
           "polyexponential": lambda n: O.get_sigmas_polyexponential(n, 0.005, 1.0),
           "vp": lambda n: O.get_sigmas_vp(n), "ve": lambda n: O.get_sigmas_ve(n, 0.005, 1.0),
           "cosine_beta": lambda n: O.cosine_beta_schedule(n)}
    for key, ref in fx.items():
        name, n = key.rsplit("_", 1)
        mine = fns[name](int(n))
        assert mine.shape == ref.shape and mine[-1] == 0
        np.testing.assert_allclose(mine, ref, rtol=3e-6, atol=1e-9, err_msg=key)


def test_trajectory_windows_match_reference():
    """The feed's restatement against TrajectorySlicerDataset items of the reference, bit for bit."""
    fx = load_golden("trajectory_windows.npz")
    sub = fx["subset"]
    obs, act, lengths = fx["observations"][sub], fx["actions"][sub], fx["lengths"][sub]
    window, glen, sep = int(fx["window"]), int(fx["future_seq_len"]), int(fx["min_future_sep"])
    table = O.window_table(lengths, window)
    np.testing.assert_array_equal(table, fx["slices"])
    for mode in ("none", "random", "tail", "seq_end"):
        out = O.slice_windows(obs, act, lengths, table, fx["ids"], goal_len=0 if mode == "none" else glen,
                              min_future_sep=sep, mode=mode, draws=fx["random::draws"])
        for key, val in out.items():
            np.testing.assert_array_equal(val, fx[f"{mode}::{key}"], err_msg=f"{mode}::{key}")
        assert ("goal_observation" in out) == (mode != "none")
