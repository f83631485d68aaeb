#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_fixtures.py

The reference (intuitive-robots/beso) is imported unmodified from /root/reference.  Five modules it
imports at module scope are absent from this image and carry no arithmetic of the hot path:
hydra / omegaconf (object construction only), torchsde / torchdiffeq (used only by
sample_dpmpp_sde / log_likelihood, outside the scope), wandb (logging).  They are replaced by the
empty stand-ins below so that the import succeeds; every number stored here is produced by the
reference's own ``DiffusionGPT`` / ``GCDenoiser`` / ``gc_sampling`` / ``ClassifierFreeSampleModel``
/ ``BesoAgent`` code running on torch CPU fp32.

Weights: the trained checkpoints are not shipped (.MISSING_LARGE_BLOBS), so weights follow the
seeded recipe ``oracle.beso_oracle.make_weights`` and are loaded with ``load_state_dict``; only the
tiny configs store their weights in the fixture, the large ones store a checksum.
Only data (inputs, outputs) is written -- no reference source in any form.
"""
import functools
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def _install_stubs():
    def instantiate(cfg, *a, **kw):
        return cfg(*a, **kw)
    hydra = types.ModuleType("hydra")
    hydra.utils = types.ModuleType("hydra.utils")
    hydra.utils.instantiate = instantiate
    hydra.main = lambda *a, **k: (lambda f: f)
    omegaconf = types.ModuleType("omegaconf")
    omegaconf.DictConfig = dict
    omegaconf.OmegaConf = type("OmegaConf", (), {})
    torchsde = types.ModuleType("torchsde")
    torchsde.BrownianTree = object
    torchdiffeq = types.ModuleType("torchdiffeq")
    torchdiffeq.odeint = None
    wandb = types.ModuleType("wandb")
    wandb.log = lambda *a, **k: None
    for name, mod in [("hydra", hydra), ("hydra.utils", hydra.utils), ("omegaconf", omegaconf),
                      ("torchsde", torchsde), ("torchdiffeq", torchdiffeq), ("wandb", wandb)]:
        sys.modules.setdefault(name, mod)


_install_stubs()
sys.path.insert(0, REF)

from beso.agents.diffusion_agents.k_diffusion.score_gpts import DiffusionGPT          # noqa: E402
from beso.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser        # noqa: E402
from beso.agents.diffusion_agents.k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel  # noqa: E402
from beso.agents.diffusion_agents.k_diffusion import gc_sampling as ref_samp          # noqa: E402
from beso.agents.diffusion_agents.k_diffusion import utils as ref_utils               # noqa: E402
from beso.agents.diffusion_agents.beso_agent import BesoAgent                         # noqa: E402
from beso.agents.input_encoders.obs_encoder import NoEncoder                          # noqa: E402
from beso.networks.scaler.scaler_class import Scaler                                  # noqa: E402

from oracle import beso_oracle as O                                                   # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def build_ref(cfg: O.ScoreGPTConfig, w, attn_pdrop=0.0, resid_pdrop=0.0, goal_drop=0.0):
    inner = functools.partial(
        DiffusionGPT, state_dim=cfg.obs_dim, device="cpu", goal_conditioned=cfg.goal_conditioned,
        action_dim=cfg.act_dim, embed_dim=cfg.embed_dim, embed_pdrob=0.0, attn_pdrop=attn_pdrop,
        resid_pdrop=resid_pdrop, n_layers=cfg.n_layers, n_heads=cfg.n_heads,
        goal_seq_len=cfg.goal_seq_len, obs_seq_len=cfg.obs_seq_len, sigma_vocab_size=3,
        time_embedding_fn=None, goal_drop=goal_drop, linear_output=cfg.linear_output)
    m = GCDenoiser(inner, sigma_data=cfg.sigma_data)
    sd = m.state_dict()
    new = {}
    for k, v in sd.items():
        if k in w:
            assert tuple(v.shape) == w[k].shape, (k, v.shape, w[k].shape)
            new[k] = torch.from_numpy(w[k].copy())
        else:
            assert k.endswith("attn.mask"), k
            new[k] = v
    assert set(w) <= set(sd), set(w) - set(sd)
    m.load_state_dict(new)
    # the oracle's canonical order must be the reference's named_parameters() order
    assert [n for n, _ in m.named_parameters()] == [n for n, _ in O.param_shapes(cfg)]
    m.eval()
    return m


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def wsum(w):
    return np.float64(sum(float(np.abs(v.astype(np.float64)).sum()) for v in w.values()))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


def sigma_vec(batch, seed, lo, hi):
    rng = np.random.Generator(np.random.PCG64([seed, 77]))
    return np.exp(rng.uniform(np.log(lo), np.log(hi), size=batch)).astype(np.float32)


# ----------------------------------------------------------------------------------------------
def fixture_forward(cfg_name, batch, seed, std, store_weights):
    cfg = O.CONFIGS[cfg_name]
    w = O.make_weights(cfg, seed=seed, std=std)
    m = build_ref(cfg, w)
    out = {"wsum": wsum(w), "seed": seed, "std": std}
    if store_weights:
        for k, v in w.items():
            out["w::" + k] = v
    ts = sorted({1, max(1, cfg.obs_seq_len // 2), cfg.obs_seq_len})
    out["ts"] = np.array(ts)
    for t in ts:
        state, goal, x = O.make_inputs(cfg, batch, seed=seed + t, t=t)
        sig = sigma_vec(batch, seed + t, 0.005, 1.0)
        out[f"t{t}::state"], out[f"t{t}::goal"], out[f"t{t}::action"], out[f"t{t}::sigma"] = state, goal, x, sig
        out[f"t{t}::denoised"] = m(T(state), T(x), T(goal), T(sig)).numpy()
        out[f"t{t}::denoised_uncond"] = m(T(state), T(x), T(goal), T(sig), uncond=True).numpy()
        out[f"t{t}::inner"] = m.inner_model(T(state), T(x), T(goal), T(sig)).numpy()
    return out


def fixture_samplers(cfg_name, batch, seed, std, specs, sigma_min, sigma_max, cond_lambda=None):
    cfg = O.CONFIGS[cfg_name]
    w = O.make_weights(cfg, seed=seed, std=std)
    m = build_ref(cfg, w)
    model = m if cond_lambda is None else ClassifierFreeSampleModel(m, cond_lambda)
    state, goal, x_t = O.make_inputs(cfg, batch, seed=seed, sigma_max=sigma_max)
    out = {"wsum": wsum(w), "seed": seed, "std": std, "state": state, "goal": goal, "x_t": x_t,
           "sigma_min": sigma_min, "sigma_max": sigma_max,
           "cond_lambda": np.float32(-1 if cond_lambda is None else cond_lambda)}
    sched = {"exponential": lambda n: ref_samp.get_sigmas_exponential(n, sigma_min, sigma_max),
             "linear": lambda n: ref_samp.get_sigmas_linear(n, sigma_min, sigma_max),
             "karras": lambda n: ref_samp.get_sigmas_karras(n, sigma_min, sigma_max, 5.0)}
    fns = {"ddim": ref_samp.sample_ddim, "euler": ref_samp.sample_euler, "heun": ref_samp.sample_heun,
           "dpmpp_2m": ref_samp.sample_dpmpp_2m, "dpm": ref_samp.sample_dpm_2, "dpmpp_2s": ref_samp.sample_dpmpp_2s}
    for sampler, n, schedule in specs:
        sig = sched[schedule](n)
        torch.manual_seed(1234)
        y = fns[sampler](model, T(state), T(x_t), T(goal), sig, disable=True)
        key = f"{sampler}_{n}_{schedule}"
        out[key + "::sigmas"] = sig.numpy()
        out[key + "::out"] = y.numpy()
    return out


def fixture_euler_ancestral(cfg_name, batch, seed, n):
    """euler_ancestral draws randn_like per step; record the draws so the noise can be injected."""
    cfg = O.CONFIGS[cfg_name]
    w = O.make_weights(cfg, seed=seed, std=0.02)
    m = build_ref(cfg, w)
    state, goal, x_t = O.make_inputs(cfg, batch, seed=seed)
    sig = ref_samp.get_sigmas_linear(n, 0.005, 1.0)
    torch.manual_seed(4321)
    y = ref_samp.sample_euler_ancestral(m, T(state), T(x_t), T(goal), sig, disable=True)
    torch.manual_seed(4321)   # replay the same stream: one randn_like per step with sigma_down > 0
    draws = []
    for i in range(n):
        sd, su = ref_samp.get_ancestral_step(sig[i], sig[i + 1])
        draws.append(torch.randn(x_t.shape).numpy() if sd > 0 else np.zeros_like(x_t))
    return {"state": state, "goal": goal, "x_t": x_t, "sigmas": sig.numpy(), "out": y.numpy(),
            "noise": np.stack(draws), "seed": seed}


def fixture_more_samplers(cfg_name, batch, seed):
    """The samplers that draw noise or choose their own steps, run on the reference with the torch CPU generator
    seeded: the host loops of the build consume the same stream, so outputs are comparable directly."""
    cfg = O.CONFIGS[cfg_name]
    w = O.make_weights(cfg, seed=seed, std=0.05)
    m = build_ref(cfg, w)
    state, goal, x_t = O.make_inputs(cfg, batch, seed=seed)
    sig = ref_samp.get_sigmas_exponential(6, 0.005, 1.0)
    out = {"state": state, "goal": goal, "x_t": x_t, "sigmas": sig.numpy(), "seed": seed, "std": 0.05}
    args = (m, T(state), T(x_t), T(goal))
    smin, smax = sig[-2].item(), sig[0].item()

    def run(key, fn):
        torch.manual_seed(999)
        y = fn()
        out[key] = (y[0] if isinstance(y, tuple) else y).numpy()
        if isinstance(y, tuple):
            out[key + "::info"] = np.array([y[1][k] for k in ("steps", "nfe", "n_accept", "n_reject")])

    run("lms", lambda: ref_samp.sample_lms(*args, sig, disable=True))
    run("ancestral", lambda: ref_samp.sample_dpm_2_ancestral(*args, sig, disable=True))
    run("dpmpp_2s_ancestral", lambda: ref_samp.sample_dpmpp_2s_ancestral(*args, sig, disable=True))
    for n in (6, 7, 8):
        run(f"dpm_fast_{n}", lambda: ref_samp.sample_dpm_fast(*args, smin, smax, n, disable=True))
    run("dpm_fast_7_eta", lambda: ref_samp.sample_dpm_fast(*args, smin, smax, 7, disable=True, eta=0.5))
    for order in (2, 3):
        run(f"dpm_adaptive_{order}", lambda: ref_samp.sample_dpm_adaptive(*args, smin, smax, disable=True, order=order,
                                                                         return_info=True))
    run("dpm_adaptive_3_eta", lambda: ref_samp.sample_dpm_adaptive(*args, smin, smax, disable=True, eta=0.3,
                                                                   return_info=True))
    run("dpm_adaptive_3_reject", lambda: ref_samp.sample_dpm_adaptive(*args, smin, smax, disable=True, h_init=4.0, rtol=0.005,
                                                                      atol=0.001, return_info=True))
    run("dpmpp_sde", lambda: ref_samp.sample_dpmpp_sde(*args, sig, disable=True,
                                                       noise_sampler=ref_samp.default_noise_sampler(T(x_t))))
    return out


def fixture_trajectory_windows(seed=7):
    """TrajectorySlicerDataset over a TrajectorySubset of a padded TensorDataset (trajectory_loader.py:44-272), the
    way get_push_train_val builds it: the window table, and items in all three future-goal modes plus the
    unconditional one.  The random mode's np.random draws are recovered by replaying the seeded stream."""
    from torch.utils.data import TensorDataset
    import importlib.util
    # beso/envs/__init__.py registers gym environments (gym is absent here); the loader module itself only needs torch,
    # so it is loaded from its file, unmodified, without running the package's __init__
    spec = importlib.util.spec_from_file_location(
        "ref_trajectory_loader", os.path.join(REF, "beso", "envs", "dataloaders", "trajectory_loader.py"))
    TL = importlib.util.module_from_spec(spec)
    import itertools
    import torch._utils
    if not hasattr(torch._utils, "_accumulate"):          # removed from torch 2.x; only random_split_traj uses it,
        torch._utils._accumulate = itertools.accumulate   # which this fixture does not call (subset built directly)
    spec.loader.exec_module(TL)

    rng = np.random.default_rng(seed)
    n, t_max, obs, act = 9, 26, 5, 3
    lengths = rng.integers(4, t_max + 1, size=n)
    lengths[0], lengths[1] = t_max, 5                     # one full-length trajectory, one shorter than the window
    observations = rng.standard_normal((n, t_max, obs)).astype(np.float32)
    actions = rng.standard_normal((n, t_max, act)).astype(np.float32)
    masks = (np.arange(t_max)[None, :] < lengths[:, None]).astype(np.float32)
    observations *= masks[..., None]
    actions *= masks[..., None]

    class Padded(TensorDataset, TL.TrajectoryDataset):
        def __init__(self):
            TensorDataset.__init__(self, T(observations), T(actions), T(masks))

        def get_seq_length(self, idx):
            return int(self.tensors[2][idx].sum().item())

        def get_all_actions(self):
            return torch.cat([self.tensors[1][i, :self.get_seq_length(i)] for i in range(len(self))])

    window, glen, sep = 6, 2, 1
    out = {"observations": observations, "actions": actions, "lengths": lengths.astype(np.int32),
           "window": window, "future_seq_len": glen, "min_future_sep": sep}
    for mode, kw in (("none", dict(future_conditional=False)),
                     ("random", dict(future_conditional=True, future_seq_len=glen, min_future_sep=sep)),
                     ("tail", dict(future_conditional=True, future_seq_len=glen, min_future_sep=sep, only_sample_tail=True)),
                     ("seq_end", dict(future_conditional=True, future_seq_len=glen, min_future_sep=sep,
                                      only_sample_seq_end=True))):
        subset = np.array([4, 0, 7, 1, 8, 2, 5], dtype=np.int64)          # a TrajectorySubset like split_traj_datasets makes
        train = TL.TrajectorySlicerDataset(TL.TrajectorySubset(Padded(), subset.tolist()), window=window, **kw)
        out["subset"] = subset
        out["slices"] = np.asarray(train.slices, dtype=np.int32)
        ids = np.random.default_rng(seed + 1).permutation(len(train))[:40]
        out["ids"] = ids.astype(np.int64)
        np.random.seed(11)
        items = [train[int(i)] for i in ids]
        out[f"{mode}::observation"] = np.stack([it["observation"].numpy() for it in items])
        out[f"{mode}::action"] = np.stack([it["action"].numpy() for it in items])
        if mode != "none":
            out[f"{mode}::goal_observation"] = np.stack([it["goal_observation"].numpy() for it in items])
        if mode == "random":
            np.random.seed(11)
            draws = []
            for i in ids:
                tr, start, end = train.slices[int(i)]
                lo, hi = end + sep, train.dataset.get_seq_length(tr) - glen
                draws.append(np.random.randint(lo, hi) - lo if lo < hi else 0)
            out["random::draws"] = np.asarray(draws, dtype=np.int64)
    return out


def fixture_cfg(cfg_name, batch, seed):
    cfg = O.CONFIGS[cfg_name]
    w = O.make_weights(cfg, seed=seed, std=0.05)
    m = build_ref(cfg, w)
    state, goal, x = O.make_inputs(cfg, batch, seed=seed)
    sig = sigma_vec(batch, seed, 0.05, 1.0)
    out = {"seed": seed, "std": 0.05, "state": state, "goal": goal, "action": x, "sigma": sig,
           "lambdas": np.array([0.0, 1.0, 1.5, 2.0], np.float32), "wsum": wsum(w)}
    for lam in [0.0, 1.0, 1.5, 2.0]:
        out[f"lam{lam}"] = ClassifierFreeSampleModel(m, lam)(T(state), T(x), T(goal), T(sig)).numpy()
    return out


def fixture_loss(cfg_name, batch, seed):
    cfg = O.CONFIGS[cfg_name]
    w = O.make_weights(cfg, seed=seed, std=0.05)
    m = build_ref(cfg, w)          # dropout 0, goal_drop 0: train() == eval() numerically
    m.train()
    state, goal, action = O.make_inputs(cfg, batch, seed=seed)
    rng = np.random.Generator(np.random.PCG64([seed, 5]))
    noise = rng.standard_normal(action.shape, dtype=np.float32)
    sig = sigma_vec(batch, seed, 0.005, 1.0)
    with torch.enable_grad():
        loss = m.loss(T(state), T(action), T(goal), T(noise), T(sig))
        loss.backward()
    out = {"seed": seed, "std": 0.05, "state": state, "goal": goal, "action": action, "noise": noise,
           "sigma": sig, "loss": np.float32(loss.item()), "wsum": wsum(w)}
    for n, p in m.named_parameters():
        g = p.grad
        out["gnorm::" + n] = np.float32(g.norm().item())
        out["gslice::" + n] = g.reshape(-1)[:8].numpy().copy()
    # rand_log_logistic with an injected uniform stream
    torch.manual_seed(7)
    u = torch.rand((64,), dtype=torch.float64)
    torch.manual_seed(7)
    s = ref_utils.rand_log_logistic((64,), loc=np.log(0.5), scale=0.5, min_value=0.005, max_value=1.0)
    out["loglogistic::u"], out["loglogistic::sigma"] = u.numpy(), s.numpy()
    return out


def fixture_schedules():
    out = {}
    for n in (1, 3, 10, 50):
        out[f"exponential_{n}"] = ref_samp.get_sigmas_exponential(n, 0.005, 1.0).numpy()
        out[f"linear_{n}"] = ref_samp.get_sigmas_linear(n, 0.005, 1.0).numpy()
        out[f"karras_{n}"] = ref_samp.get_sigmas_karras(n, 0.005, 1.0, 5.0).numpy()
        out[f"polyexponential_{n}"] = ref_samp.get_sigmas_polyexponential(n, 0.005, 1.0).numpy()
        out[f"vp_{n}"] = ref_samp.get_sigmas_vp(n).numpy()
        out[f"cosine_beta_{n}"] = ref_samp.cosine_beta_schedule(n).numpy()
        if n > 1:
            out[f"ve_{n}"] = ref_samp.get_sigmas_ve(n, 0.005, 1.0).numpy()
    return out


def _reference_agent(cfg, w, seed):
    """The reference BesoAgent around the reference model with the seeded weights, scaler fitted on seeded data."""
    model_partial = functools.partial(
        GCDenoiser,
        functools.partial(DiffusionGPT, state_dim=cfg.obs_dim, device="cpu", goal_conditioned=True,
                          action_dim=cfg.act_dim, embed_dim=cfg.embed_dim, embed_pdrob=0.0, attn_pdrop=0.0,
                          resid_pdrop=0.0, n_layers=cfg.n_layers, n_heads=cfg.n_heads,
                          goal_seq_len=cfg.goal_seq_len, obs_seq_len=cfg.obs_seq_len, sigma_vocab_size=3,
                          time_embedding_fn=None, goal_drop=0.0, linear_output=True),
        sigma_data=cfg.sigma_data)
    agent = BesoAgent(
        model=model_partial,
        input_encoder=functools.partial(NoEncoder, device="cpu", state_modality="observation",
                                        goal_modality="goal_observation"),
        optimization=lambda params: torch.optim.AdamW(params, lr=1e-4),
        device="cpu", obs_modalities=["observation"], goal_modalities=["goal_observation"],
        target_modality="action", max_train_steps=10, max_epochs=1, train_method="steps",
        eval_every_n_steps=5, use_ema=True, goal_conditioned=True, pred_last_action_only=False,
        rho=5.0, num_sampling_steps=3, lr_scheduler=lambda optimizer: torch.optim.lr_scheduler.StepLR(optimizer, 100, 0.99),
        sampler_type="ddim", sigma_data=cfg.sigma_data, sigma_min=0.005, sigma_max=1.0,
        sigma_sample_density_type="loglogistic", sigma_sample_density_mean=-0.6, sigma_sample_density_std=1.6,
        decay=0.999, update_ema_every_n_steps=1, window_size=cfg.obs_seq_len, goal_window_size=cfg.goal_seq_len)
    sd = agent.model.state_dict()
    for k in w:
        sd[k] = torch.from_numpy(w[k].copy())
    agent.model.load_state_dict(sd)
    agent.ema_helper.load_shadow_params(agent.model.get_params())
    rng = np.random.Generator(np.random.PCG64([seed, 11]))
    x_data = rng.standard_normal((50, cfg.obs_dim)).astype(np.float32)
    y_data = rng.uniform(-1, 1, size=(50, cfg.act_dim)).astype(np.float32)
    scaler = Scaler(x_data, y_data, True, "cpu")
    agent.get_scaler(scaler)
    agent.set_bounds(scaler)
    return agent, rng, x_data, y_data


def fixture_train_trace(seed=5, n_steps=4, batch=8):
    """Four BesoAgent.train_step calls of the reference (beso_agent.py:215-248: AdamW, StepLR, EMA with warm-up) on seeded
    batches; the noise and sigma each step draws are recorded by replaying the generator."""
    cfg = O.TINY
    w = O.make_weights(cfg, seed=seed, std=0.05)
    agent, rng, x_data, y_data = _reference_agent(cfg, w, seed)
    out = {"seed": seed, "std": 0.05, "x_data": x_data, "y_data": y_data, "n_steps": n_steps, "wsum": wsum(w)}
    for i in range(n_steps):
        obs = rng.standard_normal((batch, cfg.obs_seq_len, cfg.obs_dim)).astype(np.float32)
        act = rng.uniform(-1, 1, size=(batch, cfg.obs_seq_len, cfg.act_dim)).astype(np.float32)
        goal = rng.standard_normal((batch, cfg.goal_seq_len, cfg.obs_dim)).astype(np.float32)
        torch.manual_seed(300 + i)
        noise = torch.randn((batch, cfg.obs_seq_len, cfg.act_dim))                  # torch.randn_like(action), :226
        sigma = agent.make_sample_density()(shape=(batch,), device="cpu")            # :227
        torch.manual_seed(300 + i)
        with torch.enable_grad():
            loss = agent.train_step({"observation": T(obs), "action": T(act), "goal_observation": T(goal)})
        out[f"step{i}::obs"], out[f"step{i}::action"], out[f"step{i}::goal"] = obs, act, goal
        out[f"step{i}::noise"], out[f"step{i}::sigma"], out[f"step{i}::loss"] = noise.numpy(), sigma.numpy(), np.float32(loss)
    for (n, p), sh in zip(agent.model.named_parameters(), agent.ema_helper.shadow_params):
        out["final::" + n] = p.detach().reshape(-1)[:256].numpy().copy()
        out["final_norm::" + n] = np.float64(p.detach().double().norm().item())
        out["ema::" + n] = sh.detach().reshape(-1)[:256].numpy().copy()
        out["ema_norm::" + n] = np.float64(sh.detach().double().norm().item())
    return out


def fixture_agent_trace(seed=3, n_calls=6):
    """A rollout-style BesoAgent.predict trace (kitchen_workspace_manager.py:286-294 call shape)."""
    cfg = O.TINY
    w = O.make_weights(cfg, seed=seed, std=0.05)
    agent, rng, x_data, y_data = _reference_agent(cfg, w, seed)
    agent.reset()
    goal = rng.standard_normal((cfg.goal_seq_len, cfg.obs_dim)).astype(np.float32)
    out = {"seed": seed, "std": 0.05, "x_data": x_data, "y_data": y_data, "goal": goal, "n_calls": n_calls}
    for k, v in w.items():
        out["w::" + k] = v
    for c in range(n_calls):
        obs = rng.standard_normal((1, cfg.obs_dim)).astype(np.float32)
        torch.manual_seed(100 + c)
        noise = torch.randn((1, 1, cfg.act_dim))          # the draw predict() makes (beso_agent.py:357)
        torch.manual_seed(100 + c)
        pred = agent.predict({"observation": T(obs), "goal_observation": T(goal)},
                             new_sampler_type="ddim", new_sampling_steps=3, get_mean=None,
                             extra_args={}, noise_scheduler="exponential")
        out[f"call{c}::obs"], out[f"call{c}::noise"], out[f"call{c}::pred"] = obs, noise.numpy(), pred.numpy()
        out[f"call{c}::pred_shape"] = np.array(pred.shape)
    return out


def main(only=None):
    """Writes every fixture, or only the files named on the command line (regeneration is deterministic)."""
    def save_if(name, make):
        if not only or name in only:
            save(name, **make())
    save_if("tiny_forward.npz", lambda: fixture_forward("tiny", 5, seed=1, std=0.05, store_weights=True))
    save_if("tiny_mlp_head_forward.npz", lambda: fixture_forward("tiny_mlp_head", 4, seed=2, std=0.1, store_weights=True))
    save_if("tiny_nogoal_forward.npz", lambda: fixture_forward("tiny_nogoal", 4, seed=3, std=0.1, store_weights=True))
    save_if("kitchen_forward_std002.npz", lambda: fixture_forward("kitchen", 6, seed=10, std=0.02, store_weights=False))
    save_if("kitchen_forward_std008.npz", lambda: fixture_forward("kitchen", 6, seed=11, std=0.08, store_weights=False))
    save_if("block_push_forward.npz", lambda: fixture_forward("block_push", 6, seed=12, std=0.05, store_weights=False))
    save_if("long_horizon_forward.npz", lambda: fixture_forward("long_horizon", 2, seed=13, std=0.02, store_weights=False))
    save_if("kitchen_samplers.npz", lambda: fixture_samplers(
        "kitchen", 4, seed=20, std=0.04,
        specs=[("ddim", 3, "exponential"), ("ddim", 10, "exponential"), ("ddim", 3, "linear"),
               ("euler", 10, "exponential"), ("euler", 5, "karras"), ("heun", 5, "exponential"),
               ("dpmpp_2m", 5, "exponential"), ("dpm", 4, "exponential"), ("dpmpp_2s", 4, "exponential")],
        sigma_min=0.005, sigma_max=1.0))
    save_if("block_push_heun_cfg.npz", lambda: fixture_samplers(
        "block_push", 4, seed=21, std=0.05,
        specs=[("heun", 50, "exponential"), ("heun", 5, "karras"), ("ddim", 3, "exponential")],
        sigma_min=0.05, sigma_max=1.0, cond_lambda=2.0))
    save_if("long_horizon_euler.npz", lambda: fixture_samplers(
        "long_horizon", 2, seed=22, std=0.02, specs=[("euler", 10, "exponential")],
        sigma_min=0.005, sigma_max=1.0))
    save_if("tiny_euler_ancestral.npz", lambda: fixture_euler_ancestral("tiny", 4, seed=23, n=5))
    save_if("tiny_more_samplers.npz", lambda: fixture_more_samplers("tiny", 4, seed=24))
    save_if("trajectory_windows.npz", lambda: fixture_trajectory_windows())
    save_if("block_push_cfg.npz", lambda: fixture_cfg("block_push", 5, seed=30))
    save_if("tiny_loss.npz", lambda: fixture_loss("tiny", 6, seed=40))
    save_if("kitchen_loss.npz", lambda: fixture_loss("kitchen", 6, seed=41))
    save_if("block_push_loss.npz", lambda: fixture_loss("block_push", 6, seed=42))
    save_if("tiny_mlp_head_loss.npz", lambda: fixture_loss("tiny_mlp_head", 5, seed=43))
    save_if("schedules.npz", lambda: fixture_schedules())
    save_if("tiny_agent_trace.npz", lambda: fixture_agent_trace())
    save_if("tiny_train_trace.npz", lambda: fixture_train_trace())
    # BASELINE config 5: the 100-step Euler run on the long-horizon shape (gc_sampling.py:167-213)
    save_if("long_horizon_euler100.npz", lambda: fixture_samplers(
        "long_horizon", 2, seed=25, std=0.02, specs=[("euler", 100, "exponential")],
        sigma_min=0.005, sigma_max=1.0))


if __name__ == "__main__":
    main(set(sys.argv[1:]))
