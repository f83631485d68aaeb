"""Host-side mirror of the reference interface, checked on CPU.  The score network itself is
replaced by an oracle-backed callable here (the samplers and the agent accept any
``model(state, action, goal, sigma)``); the HIP network is checked in the ``-m gpu`` tests."""
import functools

import numpy as np
import pytest
import torch

from oracle import beso_oracle as O
from conftest import load_golden, weights_from_fixture, rel_err

from beso_amd.agents.diffusion_agents.k_diffusion import gc_sampling as ks
from beso_amd.agents.diffusion_agents.k_diffusion import utils as kutils
from beso_amd.agents.diffusion_agents.k_diffusion.score_gpts import DiffusionGPT
from beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser
from beso_amd.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
from beso_amd.agents.diffusion_agents.beso_agent import BesoAgent
from beso_amd.agents.input_encoders.obs_encoder import NoEncoder
from beso_amd.networks.ema_helper.ema import ExponentialMovingAverage
from beso_amd.networks.scaler.scaler_class import Scaler
from beso_amd._instantiate import instantiate


class OracleModel(torch.nn.Module):
    """tests-only stand-in for the score network: evaluates the CPU oracle."""

    def __init__(self, w, cfg, cond_lambda=None):
        super().__init__()
        self.w, self.cfg, self.cond_lambda = w, cfg, cond_lambda
        self.dummy = torch.nn.Parameter(torch.zeros(1))      # optimizers refuse an empty parameter list

    def forward(self, state, action, goal, sigma, uncond=False, **kw):
        s, a, g, sg = (t.detach().cpu().numpy() for t in (state, action, goal, sigma))
        if self.cond_lambda is None:
            out = O.denoise(self.w, self.cfg, s, a, g, sg, uncond=uncond)
        else:
            out = O.denoise_cfg(self.w, self.cfg, s, a, g, sg, self.cond_lambda)
        return torch.from_numpy(np.ascontiguousarray(out))

    def get_params(self):
        return self.parameters()


def make_module(cfg: O.ScoreGPTConfig, **kw):
    inner = functools.partial(
        DiffusionGPT, state_dim=cfg.obs_dim, device="cpu", goal_conditioned=cfg.goal_conditioned,
        action_dim=cfg.act_dim, embed_dim=cfg.embed_dim, embed_pdrob=0.0, attn_pdrop=kw.get("attn_pdrop", 0.0),
        resid_pdrop=0.0, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=cfg.goal_seq_len,
        obs_seq_len=cfg.obs_seq_len, sigma_vocab_size=3, time_embedding_fn=None,
        goal_drop=kw.get("goal_drop", 0.0), linear_output=cfg.linear_output)
    return GCDenoiser(inner, sigma_data=cfg.sigma_data)


def load_weights(module, w):
    sd = module.state_dict()
    for k, v in w.items():
        assert k in sd, k
        sd[k] = torch.from_numpy(v.copy())
    module.load_state_dict(sd)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["kitchen", "block_push", "tiny", "tiny_mlp_head", "tiny_nogoal"])
def test_module_has_the_reference_parameter_layout(name):
    cfg = O.CONFIGS[name]
    m = make_module(cfg)
    shapes = O.param_shapes(cfg)          # the reference's named_parameters() order (asserted in make_fixtures.py)
    assert [n for n, _ in m.named_parameters()] == [n for n, _ in shapes]
    assert [tuple(p.shape) for _, p in m.named_parameters()] == [s for _, s in shapes]
    masks = [k for k in m.state_dict() if k.endswith("attn.mask")]
    assert len(masks) == cfg.n_layers and len(m.state_dict()) == len(shapes) + cfg.n_layers
    assert m.state_dict()[masks[0]].shape == (1, 1, cfg.block_size, cfg.block_size)
    assert list(m.get_params()) == list(m.inner_model.parameters())


def test_autograd_comparator_matches_reference_vectors():
    """The torch-autograd comparator of the HIP training step (tests/autograd_reference.py) reproduces the reference's
    forward, loss and parameter gradients (tiny_loss.npz was produced by the reference's loss.backward())."""
    from autograd_reference import forward_autograd, loss_autograd
    fx = load_golden("tiny_loss.npz")
    cfg = O.TINY
    w = O.make_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"]))
    m = make_module(cfg)
    load_weights(m, w)
    m.train()
    T = lambda k: torch.from_numpy(fx[k].copy())        # noqa: E731
    loss = loss_autograd(m, T("state"), T("action"), T("goal"), T("noise"), T("sigma"))
    assert abs(loss.item() - float(fx["loss"])) < 2e-6 * abs(float(fx["loss"]))
    loss.backward()
    for n, p in m.named_parameters():
        assert abs(p.grad.norm().item() - float(fx["gnorm::" + n])) <= 2e-4 * float(fx["gnorm::" + n]) + 1e-9, n
        np.testing.assert_allclose(p.grad.reshape(-1)[:8].numpy(), fx["gslice::" + n], rtol=2e-3, atol=1e-7)
    fwd = load_golden("tiny_forward.npz")
    w2 = weights_from_fixture(fwd)
    load_weights(m, w2)
    m.eval()
    for t in fwd["ts"]:
        p = f"t{int(t)}::"
        x = forward_autograd(m.inner_model, *(torch.from_numpy(fwd[p + k].copy()) for k in ("state", "action", "goal", "sigma")), False)
        assert rel_err(x.detach().numpy(), fwd[p + "inner"]) < 2e-5


def test_schedules_match_reference():
    fx = load_golden("schedules.npz")
    fns = {"exponential": lambda n: ks.get_sigmas_exponential(n, 0.005, 1.0),
           "linear": lambda n: ks.get_sigmas_linear(n, 0.005, 1.0),
           "karras": lambda n: ks.get_sigmas_karras(n, 0.005, 1.0, 5.0),
           "polyexponential": lambda n: ks.get_sigmas_polyexponential(n, 0.005, 1.0),
           "vp": lambda n: ks.get_sigmas_vp(n), "ve": lambda n: ks.get_sigmas_ve(n, 0.005, 1.0),
           "cosine_beta": lambda n: ks.cosine_beta_schedule(n)}
    for key, ref in fx.items():
        name, n = key.rsplit("_", 1)
        np.testing.assert_array_equal(fns[name](int(n)).numpy(), ref, err_msg=key)


@pytest.mark.parametrize("fixture,cfg_name", [("kitchen_samplers.npz", "kitchen"),
                                              ("block_push_heun_cfg.npz", "block_push")])
def test_generic_sampler_loops_match_reference(fixture, cfg_name):
    fx = load_golden(fixture)
    cfg = O.CONFIGS[cfg_name]
    w = O.make_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"]))
    lam = float(fx["cond_lambda"])
    model = OracleModel(w, cfg, None if lam < 0 else lam)
    fns = {"ddim": ks.sample_ddim, "euler": ks.sample_euler, "heun": ks.sample_heun, "dpmpp_2m": ks.sample_dpmpp_2m,
           "dpm": ks.sample_dpm_2, "dpmpp_2s": ks.sample_dpmpp_2s}
    T = lambda k: torch.from_numpy(fx[k].copy())        # noqa: E731
    for key in sorted(k[:-5] for k in fx if k.endswith("::out")):
        n = int(key.split("_")[-2])
        if n > 10:
            continue                                    # the 50-step case is covered by the oracle test
        sampler = key[: key.index(f"_{n}_")]
        out = fns[sampler](model, T("state"), T("x_t"), T("goal"), torch.from_numpy(fx[key + "::sigmas"]), disable=True)
        assert rel_err(out.numpy(), fx[key + "::out"]) < 5e-5, key


def test_euler_ancestral_consumes_the_same_random_stream_as_the_reference():
    fx = load_golden("tiny_euler_ancestral.npz")
    cfg = O.TINY
    w = O.make_weights(cfg, seed=int(fx["seed"]), std=0.02)
    torch.manual_seed(4321)
    out = ks.sample_euler_ancestral(OracleModel(w, cfg), *(torch.from_numpy(fx[k].copy()) for k in ("state", "x_t", "goal")),
                                    torch.from_numpy(fx["sigmas"]), disable=True)
    assert rel_err(out.numpy(), fx["out"]) < 2e-5


def _more_sampler_runs(model, fx, T, noise_like):
    """key -> thunk for every entry of tests/golden/tiny_more_samplers.npz (make_fixtures.fixture_more_samplers)."""
    sig = torch.from_numpy(fx["sigmas"])
    smin, smax = sig[-2].item(), sig[0].item()
    a = lambda: (model, T("state"), T("x_t"), T("goal"))      # noqa: E731
    runs = {"lms": lambda: ks.sample_lms(*a(), sig, disable=True),
            "ancestral": lambda: ks.sample_dpm_2_ancestral(*a(), sig, disable=True),
            "dpmpp_2s_ancestral": lambda: ks.sample_dpmpp_2s_ancestral(*a(), sig, disable=True),
            "dpm_fast_7_eta": lambda: ks.sample_dpm_fast(*a(), smin, smax, 7, disable=True, eta=0.5),
            "dpm_adaptive_3_eta": lambda: ks.sample_dpm_adaptive(*a(), smin, smax, disable=True, eta=0.3, return_info=True),
            "dpm_adaptive_3_reject": lambda: ks.sample_dpm_adaptive(*a(), smin, smax, disable=True, h_init=4.0, rtol=0.005,
                                                                    atol=0.001, return_info=True),
            "dpmpp_sde": lambda: ks.sample_dpmpp_sde(*a(), sig, disable=True, noise_sampler=noise_like)}
    for n in (6, 7, 8):
        runs[f"dpm_fast_{n}"] = lambda n=n: ks.sample_dpm_fast(*a(), smin, smax, n, disable=True)
    for order in (2, 3):
        runs[f"dpm_adaptive_{order}"] = lambda order=order: ks.sample_dpm_adaptive(*a(), smin, smax, disable=True,
                                                                                  order=order, return_info=True)
    return runs


def test_stochastic_and_adaptive_samplers_match_reference():
    """lms, dpm_2_ancestral, dpmpp_2s_ancestral, dpm_fast, dpm_adaptive, dpmpp_sde (gc_sampling.py:379-468,498-699,
    739-795,855-892,971-1016) against the reference's outputs; the torch CPU generator is seeded as in the fixture
    run, so every randn the reference drew is drawn here too (including the ones DPM-Solver scales by zero)."""
    fx = load_golden("tiny_more_samplers.npz")
    cfg = O.TINY
    model = OracleModel(O.make_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"])), cfg)
    T = lambda k: torch.from_numpy(fx[k].copy())        # noqa: E731
    runs = _more_sampler_runs(model, fx, T, ks.default_noise_sampler(T("x_t")))
    assert set(runs) == {k for k in fx if "::" not in k} - {"state", "goal", "x_t", "sigmas", "seed", "std"}
    for key, fn in runs.items():
        torch.manual_seed(999)
        y = fn()
        if isinstance(y, tuple):
            y, info = y
            assert [info[k] for k in ("steps", "nfe", "n_accept", "n_reject")] == list(fx[key + "::info"]), key
        assert rel_err(y.numpy(), fx[key]) < 5e-5, key


def test_brownian_noise_sampler_is_one_path():
    """Increments over adjacent intervals add up (one Brownian path per sample), are N(0, 1) after the 1/sqrt(dt)
    normalisation, and a seed repeats them (the contract of gc_sampling.py:144-165)."""
    x = torch.zeros(4000, 3)
    s = ks.BrownianTreeNoiseSampler(x, 0.01, 1.0, seed=5)
    a, b, c = 0.9, 0.5, 0.2
    w_ab, w_bc, w_ac = s(a, b) * (a - b) ** 0.5, s(b, c) * (b - c) ** 0.5, s(a, c) * (a - c) ** 0.5
    assert torch.allclose(w_ab + w_bc, w_ac, atol=1e-5)
    assert torch.allclose(s(c, a), -s(a, c))
    for v in (s(a, b), s(b, c), s(0.7, 0.6), s(1.0, 0.01)):
        assert abs(float(v.mean())) < 0.05 and abs(float(v.std()) - 1.0) < 0.05
    assert abs(float((s(a, b) * s(b, c)).mean())) < 0.05                        # independent increments
    s2 = ks.BrownianTreeNoiseSampler(x, 0.01, 1.0, seed=5)
    assert torch.equal(s2(a, b), s(a, b))
    rows = ks.BrownianTreeNoiseSampler(x[:3], 0.01, 1.0, seed=[1, 2, 1])
    v = rows(0.8, 0.3)
    assert torch.equal(v[0], v[2]) and not torch.equal(v[0], v[1])
    with pytest.raises(NotImplementedError):
        ks.sample_dpmpp_2m_sde()


def test_cfg_wrapper_semantics():
    fx = load_golden("block_push_cfg.npz")
    cfg = O.BLOCK_PUSH
    w = O.make_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"]))
    T = lambda k: torch.from_numpy(fx[k].copy())        # noqa: E731
    for lam in fx["lambdas"]:
        m = ClassifierFreeSampleModel(OracleModel(w, cfg), float(lam))
        out = m(T("state"), T("action"), T("goal"), T("sigma"))
        assert rel_err(out.numpy(), fx[f"lam{float(lam)}"]) < 2e-5


def test_sigma_density_matches_reference():
    fx = load_golden("tiny_loss.npz")
    torch.manual_seed(7)
    s = kutils.rand_log_logistic((64,), loc=np.log(0.5), scale=0.5, min_value=0.005, max_value=1.0)
    np.testing.assert_allclose(s.numpy(), fx["loglogistic::sigma"], rtol=1e-6)
    assert kutils.append_dims(torch.ones(3), 3).shape == (3, 1, 1)
    with pytest.raises(ValueError):
        kutils.append_dims(torch.ones(3, 1, 1), 2)


def test_ema_rule():
    p = [torch.nn.Parameter(torch.ones(4, 3)), torch.nn.Parameter(torch.zeros(5))]
    ema = ExponentialMovingAverage(p, 0.999, "cpu")
    with torch.no_grad():
        p[0].mul_(3.0)
        p[1].add_(2.0)
    ema.update(p)                                        # decay = min(0.999, 2/11)
    d = 2 / 11
    np.testing.assert_allclose(ema.shadow_params[0].numpy(), 1 - (1 - d) * (1 - 3.0), rtol=1e-6)
    np.testing.assert_allclose(ema.shadow_params[1].numpy(), 0 - (1 - d) * (0 - 2.0), rtol=1e-6)
    ema.store(p)
    ema.copy_to(p)
    np.testing.assert_allclose(p[1].detach().numpy(), ema.shadow_params[1].numpy())
    ema.restore(p)
    np.testing.assert_allclose(p[1].detach().numpy(), 2.0)
    with pytest.raises(ValueError):
        ExponentialMovingAverage(p, 1.5)


def test_instantiate_shim():
    enc = instantiate({"_target_": "beso_amd.agents.input_encoders.obs_encoder.NoEncoder", "_recursive_": False,
                       "device": "cpu", "state_modality": "observation", "goal_modality": "goal_observation"})
    assert isinstance(enc, NoEncoder)
    assert instantiate(functools.partial(dict, a=1), b=2) == {"a": 1, "b": 2}
    with pytest.raises(TypeError):
        instantiate(3)


# ------------------------------------------------------------------------------------------------
def build_agent(cfg, model_factory, device="cpu", sampler="ddim", lr=1e-4):
    return BesoAgent(
        model=model_factory,
        input_encoder=functools.partial(NoEncoder, device=device, state_modality="observation",
                                        goal_modality="goal_observation"),
        optimization=lambda params: torch.optim.AdamW(params, lr=lr),
        device=device, obs_modalities=["observation"], goal_modalities=["goal_observation"],
        target_modality="action", max_train_steps=10, max_epochs=1, train_method="steps", eval_every_n_steps=5,
        use_ema=True, goal_conditioned=True, pred_last_action_only=False, rho=5.0, num_sampling_steps=3,
        lr_scheduler=lambda optimizer: torch.optim.lr_scheduler.StepLR(optimizer, 100, 0.99),
        sampler_type=sampler, sigma_data=cfg.sigma_data, sigma_min=0.005, sigma_max=1.0,
        sigma_sample_density_type="loglogistic", sigma_sample_density_mean=-0.6, sigma_sample_density_std=1.6,
        decay=0.999, update_ema_every_n_steps=1, window_size=cfg.obs_seq_len, goal_window_size=cfg.goal_seq_len)


def run_agent_trace(agent, fx, device="cpu"):
    """Replays tiny_agent_trace.npz: returns max relative error of the predicted actions."""
    agent.get_scaler(Scaler(fx["x_data"], fx["y_data"], True, device))
    agent.set_bounds(agent.scaler)
    agent.reset()
    goal = torch.from_numpy(fx["goal"].copy())
    worst = 0.0
    for c in range(int(fx["n_calls"])):
        noise = torch.from_numpy(fx[f"call{c}::noise"])
        real_randn = torch.randn
        try:   # inject the x_T draw the reference made (beso_agent.py:357); RNG parity is by injection
            torch.randn = lambda *a, **k: noise.to(k.get("device", "cpu")).clone()
            pred = agent.predict({"observation": torch.from_numpy(fx[f"call{c}::obs"].copy()),
                                  "goal_observation": goal}, new_sampler_type="ddim", new_sampling_steps=3,
                                 get_mean=None, extra_args={}, noise_scheduler="exponential")
        finally:
            torch.randn = real_randn
        assert tuple(pred.shape) == tuple(fx[f"call{c}::pred_shape"]), (c, pred.shape)   # [1,1,act] then [1,act]
        worst = max(worst, rel_err(pred.detach().cpu().numpy(), fx[f"call{c}::pred"]))
    return worst


def test_agent_predict_trace_matches_reference():
    fx = load_golden("tiny_agent_trace.npz")
    cfg = O.TINY
    w = weights_from_fixture(fx)
    agent = build_agent(cfg, lambda: OracleModel(w, cfg))
    assert run_agent_trace(agent, fx) < 5e-5
    assert len(agent.obs_context) == cfg.obs_seq_len and len(agent.action_context) == cfg.obs_seq_len - 1
    agent.reset()
    assert len(agent.obs_context) == 0 and len(agent.action_context) == 0


def test_processed_goal_is_reused_only_while_nothing_it_depends_on_changed():
    """process_batch(predict=True) keeps the processed goal with the tensor it came from (a rollout passes the same goal object
    step after step): the same object -> the same processed tensor; a write into the goal, another goal object, another
    scaler or new scaler statistics -> processed afresh, equal to what an agent without a cache computes."""
    cfg = O.TINY
    rng = np.random.default_rng(3)
    mk = lambda: Scaler(rng.standard_normal((64, cfg.obs_dim)).astype(np.float32),            # noqa: E731
                        rng.standard_normal((64, cfg.act_dim)).astype(np.float32), True, "cpu")
    agent = build_agent(cfg, lambda: OracleModel(O.make_weights(cfg), cfg))
    agent.get_scaler(mk())
    goal = torch.randn(cfg.goal_seq_len, cfg.obs_dim)
    obs = lambda: torch.randn(1, cfg.obs_dim)                                                  # noqa: E731
    fresh = lambda g: agent.scaler.scale_input(g.clone())                                      # noqa: E731
    _, g1, _ = agent.process_batch({"observation": obs(), "goal_observation": goal}, predict=True)
    _, g2, _ = agent.process_batch({"observation": obs(), "goal_observation": goal}, predict=True)
    assert g2 is g1 and torch.equal(g1, fresh(goal))
    goal[0, 0] += 1.0                                               # in-place write: the version counter moves
    _, g3, _ = agent.process_batch({"observation": obs(), "goal_observation": goal}, predict=True)
    assert g3 is not g1 and torch.equal(g3, fresh(goal))
    other = goal.clone()
    _, g4, _ = agent.process_batch({"observation": obs(), "goal_observation": other}, predict=True)
    assert g4 is not g3 and torch.equal(g4, g3)
    agent.get_scaler(mk())                                          # another scaler
    _, g5, _ = agent.process_batch({"observation": obs(), "goal_observation": other}, predict=True)
    assert g5 is not g4 and torch.equal(g5, fresh(other)) and not torch.equal(g5, g4)
    agent.scaler.x_mean.add_(0.5)                                   # new statistics in the same scaler
    _, g6, _ = agent.process_batch({"observation": obs(), "goal_observation": other}, predict=True)
    assert g6 is not g5 and torch.equal(g6, fresh(other))
    # a training batch (it holds the target) never touches the cache
    state, action, g7 = agent.process_batch({"observation": torch.randn(2, 3, cfg.obs_dim), "action": torch.randn(2, 3, cfg.act_dim),
                                             "goal_observation": other}, predict=False)
    assert g7 is not g6 and torch.equal(g7, g6)


def test_agent_surface_and_error_conventions():
    cfg = O.TINY
    agent = build_agent(cfg, lambda: OracleModel(O.make_weights(cfg), cfg))
    with pytest.raises(ValueError, match="sampler"):
        agent.sample_loop(torch.tensor([1.0, 0.0]), torch.zeros(1, 1, 3), torch.zeros(1, 1, 7), torch.zeros(1, 2, 7), "nope")
    with pytest.raises(ValueError):
        agent.get_noise_schedule(3, "nope")
    with pytest.raises(KeyError):      # non-empty extra_args must carry both keys (beso_agent.py:408-410)
        agent.sample_loop(torch.tensor([1.0, 0.0]), torch.zeros(1, 1, 3), torch.zeros(1, 1, 7), torch.zeros(1, 2, 7),
                          "ddim", {"s_churn": 1})
    for kind in ("loglogistic", "lognormal", "loguniform", "uniform", "v-diffusion"):
        agent.sigma_sample_density_type = kind
        s = agent.make_sample_density()(shape=(16,), device="cpu")
        assert s.shape == (16,) and bool((s > 0).all())
    agent.sigma_sample_density_type = "nope"
    with pytest.raises(ValueError):
        agent.make_sample_density()
    for name in ("karras", "exponential", "vp", "linear", "cosine_beta", "ve", "iddpm"):
        sig = agent.get_noise_schedule(5, name)
        assert sig.shape == (6,) and float(sig[-1]) == 0.0


def test_train_step_and_checkpoint_roundtrip(tmp_path, autograd_training):
    """train_step (host logic on CPU, the loss through the autograd comparator), EMA update, store_model_weights / load_pretrained_model formats
    (model_state_dict.pth = EMA weights, non_ema_model_state_dict.pth = raw: beso_agent.py:466-476)."""
    cfg = O.TINY
    agent = build_agent(cfg, lambda: make_module(cfg, attn_pdrop=0.1, goal_drop=0.1))
    rng = np.random.default_rng(0)
    x_data = rng.standard_normal((40, cfg.obs_dim)).astype(np.float32)
    y_data = rng.uniform(-1, 1, (40, cfg.act_dim)).astype(np.float32)
    agent.get_scaler(Scaler(x_data, y_data, True, "cpu"))
    batch = {"observation": torch.randn(8, cfg.obs_seq_len, cfg.obs_dim),
             "goal_observation": torch.randn(8, cfg.goal_seq_len, cfg.obs_dim),
             "action": torch.rand(8, cfg.obs_seq_len, cfg.act_dim) * 2 - 1}
    before = [p.detach().clone() for p in agent.model.parameters()]
    l0 = agent.train_step(batch)
    assert np.isfinite(l0) and agent.steps == 1 and agent.ema_helper.num_updates == 1
    assert any(not torch.equal(a, b) for a, b in zip(before, agent.model.parameters()))
    agent.working_dir = str(tmp_path)
    agent.store_model_weights(str(tmp_path))
    ema_sd = torch.load(tmp_path / "model_state_dict.pth")
    raw_sd = torch.load(tmp_path / "non_ema_model_state_dict.pth")
    assert list(ema_sd) == list(agent.model.state_dict()) == list(raw_sd)
    k = "inner_model.tok_emb.weight"
    assert torch.equal(raw_sd[k], agent.model.state_dict()[k])
    assert torch.allclose(ema_sd[k], agent.ema_helper.shadow_params[1]) and not torch.equal(ema_sd[k], raw_sd[k])
    agent2 = build_agent(cfg, lambda: make_module(cfg))
    agent2.load_pretrained_model(str(tmp_path))
    assert torch.equal(agent2.model.state_dict()[k], ema_sd[k])
    assert torch.equal(agent2.ema_helper.shadow_params[1], ema_sd[k])


def test_epoch_mode_keeps_the_reference_loop_cadence(autograd_training, tmp_path):
    """train_agent_on_epochs (beso_agent.py:129-175): per training batch the step counter advances TWICE (train_step's own
    increment + the loop's, :152) and the LR scheduler gets an extra step whenever the counter reaches a multiple of
    eval_every_n_steps (:153-154); the test MSE handed to early stopping is the last test batch's."""
    cfg = O.TINY
    agent = build_agent(cfg, lambda: make_module(cfg))
    agent.eval_every_n_steps = 4
    rng = np.random.default_rng(0)
    agent.get_scaler(Scaler(rng.standard_normal((40, cfg.obs_dim)).astype(np.float32),
                            rng.uniform(-1, 1, (40, cfg.act_dim)).astype(np.float32), True, "cpu"))
    agent.working_dir = str(tmp_path)
    mk = lambda: {"observation": torch.randn(4, cfg.obs_seq_len, cfg.obs_dim),         # noqa: E731
                  "goal_observation": torch.randn(4, cfg.goal_seq_len, cfg.obs_dim),
                  "action": torch.rand(4, cfg.obs_seq_len, cfg.act_dim) * 2 - 1}
    train, test = [mk() for _ in range(5)], [mk() for _ in range(2)]
    seen = []
    agent.evaluate = lambda b: (seen.append(len(seen)), 0.5 + len(seen))[1]       # (sampling is GPU-only: stub the test MSE)
    stops = []
    real_stop = agent.early_stopping
    agent.early_stopping = lambda best, mse, patience, epochs: (stops.append(mse), real_stop(best, mse, patience, epochs))[1]
    sched_steps = []
    real_sched = agent.lr_scheduler.step
    agent.lr_scheduler.step = lambda *a, **k: (sched_steps.append(agent.steps), real_sched(*a, **k))[1]
    agent.train_agent_on_epochs(train, test, 1)
    assert agent.steps == 10 and agent.ema_helper.num_updates == 5          # two counts per batch, one EMA update per train_step
    # 5 scheduler steps from train_step (at steps 1, 3, 5, 7, 9) + extra ones when the doubled counter hits 4 and 8
    assert sched_steps == [1, 3, 4, 5, 7, 8, 9]
    assert len(seen) == 2 and stops == [2.5]                                # the LAST test batch's value
    assert (tmp_path / "model_state_dict.pth").exists()


def test_fused_optimizer_is_not_used_on_cpu():
    """maybe_fuse only replaces torch Adam / AdamW over HIP fp32 parameters; on CPU (no kernel: there is no
    CPU implementation) and for any other optimizer the object comes back unchanged."""
    from beso_amd.optim import FusedAdam, maybe_fuse
    p = [torch.nn.Parameter(torch.randn(4, 3))]
    for opt in (torch.optim.AdamW(p, lr=1e-3), torch.optim.Adam(p, lr=1e-3), torch.optim.SGD(p, lr=1e-3)):
        assert maybe_fuse(opt) is opt
    f = FusedAdam(p, lr=1e-3, decoupled_weight_decay=True)
    assert f.param_groups[0]["lr"] == 1e-3 and f.param_groups[0]["betas"] == (0.9, 0.999)
    torch.optim.lr_scheduler.StepLR(f, 100, 0.99)           # LR schedulers attach to it
    p[0].grad = torch.ones_like(p[0])
    with pytest.raises(ValueError):
        f.step()                                              # CPU parameters: refused, no fallback
    with pytest.raises(ValueError):
        FusedAdam(p, lr=-1.0)


def test_device_prefetcher_passthrough_and_order():
    """DevicePrefetcher yields every batch once, in order, with non-tensor entries untouched; on a CPU device it
    hands the loader's own objects through."""
    from beso_amd.data import DevicePrefetcher
    batches = [{"observation": torch.full((2, 3), float(i)), "action": torch.full((2, 1), float(-i)), "tag": f"b{i}"}
               for i in range(5)]
    got = list(DevicePrefetcher(batches, "cpu", depth=2))
    assert [b["tag"] for b in got] == [f"b{i}" for i in range(5)]
    assert all(g is b for g, b in zip(got, batches))
    assert len(DevicePrefetcher(batches, "cpu")) == 5
    assert list(DevicePrefetcher([], "cpu")) == []


def test_hip_training_step_is_gpu_only_and_shape_gated():
    """The HIP training step binds to HIP parameters only and there is nothing behind it: on CPU GCDenoiser.loss raises
    (no torch-op evaluation of the network in the product), as does a direct forward under autograd; both action heads
    and all three dropouts are covered by the kernels, an embedding width that is not a multiple of 8 is not."""
    from beso_amd.training import HipTrainStep
    cfg = O.TINY
    from beso_amd.agents.diffusion_agents.k_diffusion.score_gpts import DiffusionGPT
    kw = dict(state_dim=cfg.obs_dim, device="cpu", goal_conditioned=True, action_dim=cfg.act_dim, embed_dim=cfg.embed_dim,
              attn_pdrop=0.1, resid_pdrop=0.1, n_layers=cfg.n_layers, n_heads=cfg.n_heads, goal_seq_len=cfg.goal_seq_len,
              obs_seq_len=cfg.obs_seq_len)
    assert HipTrainStep.supported(DiffusionGPT(embed_pdrob=0.0, linear_output=True, **kw))
    assert HipTrainStep.supported(DiffusionGPT(embed_pdrob=0.0, linear_output=False, **kw))
    assert HipTrainStep.supported(DiffusionGPT(embed_pdrob=0.1, linear_output=True, **kw))
    kw12 = dict(kw, embed_dim=36)
    assert not HipTrainStep.supported(DiffusionGPT(embed_pdrob=0.0, linear_output=True, **kw12))
    from beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser
    den = GCDenoiser(DiffusionGPT(embed_pdrob=0.0, linear_output=True, **kw), sigma_data=0.5).train()
    B = 3
    state, action = torch.randn(B, cfg.obs_seq_len, cfg.obs_dim), torch.randn(B, cfg.obs_seq_len, cfg.act_dim)
    goal, noise, sigma = torch.randn(B, cfg.goal_seq_len, cfg.obs_dim), torch.randn(B, cfg.obs_seq_len, cfg.act_dim), torch.rand(B) + 0.1
    assert den.hip_train_step(state, action, goal, noise, sigma) is None
    with pytest.raises(ValueError, match="not on the GPU"):
        den.loss(state, action, goal, noise, sigma)
    with pytest.raises(ValueError, match="unsupported keyword"):
        den.loss(state, action, goal, noise, sigma, some_future_flag=True)
    den36 = GCDenoiser(DiffusionGPT(embed_pdrob=0.0, linear_output=True, **kw12), sigma_data=0.5).train()
    with pytest.raises(ValueError, match="multiple of 8"):
        den36.loss(state, action, goal, noise, sigma)
    with pytest.raises(RuntimeError, match="no_grad"):
        den.inner_model(state, action, goal, sigma)          # a differentiable forward does not exist
    assert all(p.grad is None for p in den.parameters())


def test_synthetic_workload_recipe_matches_the_oracles():
    """beso_amd/synthetic.py (what bench.py and the tools time) and the oracle's own copy of the recipe: same shapes,
    same seeded weights and inputs, same FLOP count."""
    from beso_amd import synthetic as S
    assert set(S.SHAPES) == set(O.CONFIGS)
    for name, shape in S.SHAPES.items():
        cfg = O.CONFIGS[name]
        assert shape.as_dict() == cfg.as_dict() and shape.flops_per_sample() == cfg.flops_per_sample()
        assert (shape.G, shape.block_size, shape.seq_size) == (cfg.G, cfg.block_size, cfg.seq_size)
        if name in ("kitchen", "long_horizon"):
            continue                                            # (same code path, 10 M parameters each)
        a, b = S.make_weights(shape, seed=3, std=0.05), O.make_weights(cfg, seed=3, std=0.05)
        assert list(a) == list(b) and all(np.array_equal(a[k], b[k]) for k in a)
        assert all(np.array_equal(x, y) for x, y in zip(S.make_inputs(shape, 5, seed=2, t=1), O.make_inputs(cfg, 5, seed=2, t=1)))


def replay_train_trace(agent, fx, device="cpu"):
    """Replays tests/golden/tiny_train_trace.npz (four train_step calls of the REFERENCE agent with the noise and sigma
    it drew injected): returns (per-step relative loss error, worst parameter error, worst EMA error), parameter and
    EMA errors relative to the largest entry of the stored slice / the stored norm."""
    agent.get_scaler(Scaler(fx["x_data"], fx["y_data"], True, device))
    agent.set_bounds(agent.scaler)
    loss_err = []
    real_randn_like, real_density = torch.randn_like, agent.make_sample_density
    try:
        for i in range(int(fx["n_steps"])):
            noise = torch.from_numpy(fx[f"step{i}::noise"]).to(device)
            sigma = torch.from_numpy(fx[f"step{i}::sigma"]).to(device)
            torch.randn_like = lambda t, *a, **k: noise.clone()
            agent.make_sample_density = lambda: (lambda shape, device: sigma.clone())
            loss = agent.train_step({"observation": torch.from_numpy(fx[f"step{i}::obs"].copy()),
                                     "action": torch.from_numpy(fx[f"step{i}::action"].copy()),
                                     "goal_observation": torch.from_numpy(fx[f"step{i}::goal"].copy())})
            ref = float(fx[f"step{i}::loss"])
            loss_err.append(abs(loss - ref) / abs(ref))
    finally:
        torch.randn_like, agent.make_sample_density = real_randn_like, real_density
    p_err = e_err = 0.0
    for (n, p), sh in zip(agent.model.named_parameters(), agent.ema_helper.shadow_params):
        for got, key, nkey in ((p.detach(), "final::" + n, "final_norm::" + n), (sh.detach(), "ema::" + n, "ema_norm::" + n)):
            ref = fx[key]
            err = float(np.abs(got.reshape(-1)[:256].cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-6))
            err = max(err, abs(float(got.double().norm()) - float(fx[nkey])) / max(float(fx[nkey]), 1e-6))
            if key.startswith("final"):
                p_err = max(p_err, err)
            else:
                e_err = max(e_err, err)
    return loss_err, p_err, e_err


def test_train_steps_match_the_reference_agent_trace(autograd_training):
    """BesoAgent.train_step x 4 on CPU (the loss through the autograd comparator, torch AdamW, StepLR, the EMA helper with its warm-up rule)
    against the trace the reference's own agent produced."""
    fx = load_golden("tiny_train_trace.npz")
    cfg = O.TINY
    w = O.make_weights(cfg, seed=int(fx["seed"]), std=float(fx["std"]))
    def model():
        m = make_module(cfg)
        load_weights(m, w)
        return m

    agent = build_agent(cfg, model, device="cpu")
    agent.ema_helper.load_shadow_params(agent.model.get_params())
    loss_err, p_err, e_err = replay_train_trace(agent, fx)
    assert max(loss_err) < 2e-5, loss_err
    # AdamW's first steps move every weight by ~lr whatever the gradient's size: last-bit differences in tiny
    # gradients flip to O(lr) differences, so parameters are compared at a few lr relative to their largest entry
    assert p_err < 5e-3 and e_err < 5e-3, (p_err, e_err)


def test_feed_window_table_and_cpu_refusal():
    """The feed's window table equals the reference slicer's; a CPU device is refused (the dataset lives in HBM)."""
    from beso_amd.data.trajectory_feed import DeviceTrajectoryFeed, window_table
    fx = load_golden("trajectory_windows.npz")
    lengths = fx["lengths"][fx["subset"]]
    traj, start = window_table(lengths, int(fx["window"]))
    np.testing.assert_array_equal(np.stack([traj, start, start + int(fx["window"])], 1), fx["slices"])
    np.testing.assert_array_equal(np.stack([traj, start, start + int(fx["window"])], 1), O.window_table(lengths, int(fx["window"])))
    assert window_table([3, 2], 5)[0].size == 0
    with pytest.raises((ValueError, RuntimeError)):
        DeviceTrajectoryFeed(fx["observations"], fx["actions"], fx["lengths"], 6, 8, "cpu")


def _shipped_config(dev, task):
    """The nested `_target_` config the reference ships (keys of configs/agents/beso_{kitchen,block_push}.yaml,
    agents/model/diffusion_gpt.yaml, agents/input_encoder/no_encoder.yaml, interpolations resolved from
    franka_kitchen_main_config.yaml:23-66 / block_push_main_config.yaml:27-67), `beso.` -> `beso_amd.` in the targets."""
    k = task == "kitchen"
    cfg = {
        "_target_": "beso_amd.agents.diffusion_agents.beso_agent.BesoAgent", "_recursive_": False,
        "model": {
            "_target_": "beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers.GCDenoiser", "_recursive_": False,
            "sigma_data": 0.5,
            "inner_model": {
                "_target_": "beso_amd.agents.diffusion_agents.k_diffusion.score_gpts.DiffusionGPT",
                "state_dim": 30 if k else 10, "action_dim": 9 if k else 2, "goal_conditioned": True, "embed_dim": 360 if k else 240,
                "n_layers": 6 if k else 4, "goal_seq_len": 2 if k else 1,
                "obs_seq_len": 4 if k else 5, "sigma_vocab_size": 3, "embed_pdrob": 0, "goal_drop": 0.1, "attn_pdrop": 0.3 if k else 0.05,
                "resid_pdrop": 0 if k else 0.05,
                "time_embedding_fn": {"_target_": "beso_amd.agents.diffusion_agents.k_diffusion.utils.return_time_sigma_embedding_model",
                                      "embedding_type": "Linear", "time_embed_dim": 360 if k else 240, "device": dev},
                "n_heads": 6 if k else 12, "device": dev, "linear_output": True}},
        "input_encoder": {"_target_": "beso_amd.agents.input_encoders.obs_encoder.NoEncoder", "device": dev,
                          "state_modality": "observation", "goal_modality": "goal_observation"},
        "optimization": ({"_target_": "torch.optim.AdamW", "lr": 1e-4, "betas": [0.9, 0.999]} if k
                         else {"_target_": "torch.optim.Adam", "lr": 1e-4}),
        "lr_scheduler": {"_target_": "torch.optim.lr_scheduler.StepLR", "step_size": 100, "gamma": 0.99},
        "obs_modalities": ["observation"], "goal_modalities": ["goal_observation"], "target_modality": "action",
        "train_method": "steps", "max_epochs": 500 if k else 100, "goal_conditioned": True, "pred_last_action_only": False,
        "eval_every_n_steps": 4000, "max_train_steps": 40000 if k else 60000, "num_sampling_steps": 3, "sampler_type": "ddim", "sigma_data": 0.5,
        "rho": 5.0, "sigma_min": 0.005 if k else 0.05, "sigma_max": 1, "sigma_sample_density_type": "loglogistic",
        "sigma_sample_density_mean": -0.6, "sigma_sample_density_std": 1.6, "use_ema": True, "decay": 0.999, "device": dev,
        "update_ema_every_n_steps": 1, "goal_window_size": 2 if k else 1, "window_size": 4 if k else 5, "patience": 80}
    return cfg


def test_shipped_kitchen_config_surface():
    """The agent built from the kitchen config (device cpu: no GPU in this suite): 9,381,249 parameters, AdamW,
    StepLR, the contexts sized by the window."""
    dev = "cpu"
    agent = instantiate(_shipped_config(dev, "kitchen"))
    assert isinstance(agent, BesoAgent) and isinstance(agent.input_encoder, NoEncoder)
    assert sum(p.numel() for p in agent.model.parameters()) == 9_381_249
    inner = agent.model.inner_model
    assert (inner.embed_dim, inner.goal_seq_len, inner.obs_seq_len) == (360, 2, 4) and inner._pdrops == (0, 0.3, 0)
    assert type(agent.optimizer).__name__ in ("AdamW", "FusedAdam") and agent.optimizer.param_groups[0]["lr"] == 1e-4
    assert isinstance(agent.lr_scheduler, torch.optim.lr_scheduler.StepLR) and agent.lr_scheduler.step_size == 100
    assert agent.obs_context.maxlen == 4 and agent.goal_context.maxlen == 2 and agent.action_context.maxlen == 3
    assert (agent.sampler_type, agent.num_sampling_steps, agent.sigma_min, agent.sigma_max) == ("ddim", 3, 0.005, 1)
    sig = agent.make_sample_density()(shape=(64,), device=dev)
    assert sig.shape == (64,) and float(sig.min()) >= 0.005 and float(sig.max()) <= 1.0
    np.testing.assert_allclose(agent.get_noise_schedule(3, "exponential").numpy(),
                               ks.get_sigmas_exponential(3, 0.005, 1, dev).numpy())


def test_shipped_block_push_config_surface():
    dev = "cpu"
    agent = instantiate(_shipped_config(dev, "block_push"))
    inner = agent.model.inner_model
    assert (inner.embed_dim, inner.n_heads, inner.goal_seq_len, inner.obs_seq_len) == (240, 12, 1, 5)
    assert inner._pdrops == (0, 0.05, 0.05) and len(inner.blocks) == 4
    assert type(agent.optimizer).__name__ in ("Adam", "FusedAdam") and agent.sigma_min == 0.05
    assert sum(p.numel() for p in agent.model.parameters()) == sum(int(np.prod(sh)) for _, sh in O.param_shapes(O.BLOCK_PUSH))
    assert agent.obs_context.maxlen == 5 and agent.goal_context.maxlen == 1


def test_numa_pinning_reads_the_topology_and_is_best_effort(tmp_path, monkeypatch):
    """pin_to_gpu_numa_node: the GPU's PCI function -> numa_node -> that node's cpulist -> sched_setaffinity, from a fake
    sysfs tree; an unreadable topology changes nothing and returns None."""
    import os
    import types
    from beso_amd import distributed as bdist
    sysfs = tmp_path / "sys"
    dev = sysfs / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = sysfs / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    have = sorted(os.sched_getaffinity(0))
    (node / "cpulist").write_text(f"{have[0]}-{have[0]},{have[-1]}\n")
    monkeypatch.setattr(torch.cuda, "get_device_properties",
                        lambda i: types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0xc1, pci_device_id=0))
    pinned = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: pinned.append(sorted(cpus)))
    assert bdist.pin_to_gpu_numa_node(0, sysfs=str(sysfs)) == 1
    assert pinned == [sorted({have[0], have[-1]})]
    assert bdist.pin_to_gpu_numa_node(0, sysfs=str(tmp_path / "nothing")) is None and len(pinned) == 1
