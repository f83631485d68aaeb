"""CPU oracle for the BESO score-denoising hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm (intuitive-robots/beso) for the path
GCDenoiser -> DiffusionGPT -> gc_sampling loop.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing
under ``beso_amd/`` imports it, and the product path raises if the HIP library is missing.

Parity pin: the reference ships no tests or golden vectors for this path, so
the oracle is pinned against outputs of the reference itself: ``tests/golden/make_fixtures.py``
imports ``/root/reference`` in the build container and stores input/output vectors under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file against them.

All citations are file:line relative to the reference checkout.
Arithmetic is fp32 by default (the reference is fp32 throughout); pass ``dtype=np.float64`` to use
the oracle as a higher-precision arbiter.

Weights are a flat dict keyed exactly like ``GCDenoiser.state_dict()`` ("inner_model.tok_emb.weight",
"inner_model.blocks.0.attn.key.weight", ...), Linear weights in torch layout [out, in].
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Callable, Dict, Optional

import numpy as np
from scipy.special import erf as _erf


# --------------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ScoreGPTConfig:
    """Hyper-parameters of DiffusionGPT + GCDenoiser (score_gpts.py:121-139, score_wrappers.py:26)."""
    obs_dim: int
    act_dim: int
    embed_dim: int
    n_layers: int
    n_heads: int
    goal_seq_len: int
    obs_seq_len: int
    goal_conditioned: bool = True
    linear_output: bool = True
    sigma_data: float = 0.5

    @property
    def G(self) -> int:
        return self.goal_seq_len if self.goal_conditioned else 0   # score_gpts.py:143-144

    @property
    def block_size(self) -> int:
        return self.G + 2 * self.obs_seq_len + 1                    # score_gpts.py:148

    @property
    def seq_size(self) -> int:
        return self.G + self.obs_seq_len + 1                        # score_gpts.py:150

    def flops_per_sample(self, t: Optional[int] = None) -> int:
        """Algorithmic FLOPs of one score-net forward for one sample."""
        t = self.obs_seq_len if t is None else t
        D, L, G = self.embed_dim, self.n_layers, self.G
        T = 1 + G + 2 * t
        return (L * (24 * T * D * D + 4 * T * T * D)
                + 2 * D * (t * self.obs_dim + G * self.obs_dim + t * self.act_dim + 1)
                + 2 * t * D * self.act_dim)

    def as_dict(self):
        return asdict(self)


KITCHEN = ScoreGPTConfig(obs_dim=30, act_dim=9, embed_dim=360, n_layers=6, n_heads=6,
                         goal_seq_len=2, obs_seq_len=4, sigma_data=0.5)
BLOCK_PUSH = ScoreGPTConfig(obs_dim=10, act_dim=2, embed_dim=240, n_layers=4, n_heads=12,
                            goal_seq_len=1, obs_seq_len=5, sigma_data=0.5)
LONG_HORIZON = ScoreGPTConfig(obs_dim=30, act_dim=9, embed_dim=512, n_layers=6, n_heads=8,
                              goal_seq_len=2, obs_seq_len=32, sigma_data=0.5)
TINY = ScoreGPTConfig(obs_dim=7, act_dim=3, embed_dim=48, n_layers=2, n_heads=6,
                      goal_seq_len=2, obs_seq_len=3, sigma_data=0.5)
TINY_MLP_HEAD = ScoreGPTConfig(obs_dim=5, act_dim=2, embed_dim=32, n_layers=1, n_heads=4,
                               goal_seq_len=1, obs_seq_len=2, linear_output=False, sigma_data=1.0)
TINY_NOGOAL = ScoreGPTConfig(obs_dim=6, act_dim=4, embed_dim=40, n_layers=2, n_heads=5,
                             goal_seq_len=2, obs_seq_len=3, goal_conditioned=False, sigma_data=0.5)

CONFIGS = {"kitchen": KITCHEN, "block_push": BLOCK_PUSH, "long_horizon": LONG_HORIZON,
           "tiny": TINY, "tiny_mlp_head": TINY_MLP_HEAD, "tiny_nogoal": TINY_NOGOAL}


# --------------------------------------------------------------------------------------------
# weights: names, shapes, seeded recipe
# --------------------------------------------------------------------------------------------
def param_shapes(cfg: ScoreGPTConfig) -> "list[tuple[str, tuple]]":
    """(name, shape) in the order of the reference module's ``named_parameters()``
    (construction order in score_gpts.py:149-191: tok_emb, pos_emb(*), blocks, ln_f, sigma_emb,
    action_emb, action_pred; torch lists direct parameters first, so pos_emb leads)."""
    D, P = cfg.embed_dim, "inner_model."
    out = [(P + "pos_emb", (1, cfg.seq_size, D)),
           (P + "tok_emb.weight", (D, cfg.obs_dim)), (P + "tok_emb.bias", (D,))]
    for i in range(cfg.n_layers):
        b = f"{P}blocks.{i}."
        out += [(b + "ln1.weight", (D,)), (b + "ln1.bias", (D,)),
                (b + "ln2.weight", (D,)), (b + "ln2.bias", (D,))]
        for n in ("key", "query", "value", "proj"):
            out += [(f"{b}attn.{n}.weight", (D, D)), (f"{b}attn.{n}.bias", (D,))]
        out += [(b + "mlp.0.weight", (4 * D, D)), (b + "mlp.0.bias", (4 * D,)),
                (b + "mlp.2.weight", (D, 4 * D)), (b + "mlp.2.bias", (D,))]
    out += [(P + "ln_f.weight", (D,)), (P + "ln_f.bias", (D,)),
            (P + "sigma_emb.weight", (D, 1)), (P + "sigma_emb.bias", (D,)),
            (P + "action_emb.weight", (D, cfg.act_dim)), (P + "action_emb.bias", (D,))]
    if cfg.linear_output:
        out += [(P + "action_pred.weight", (cfg.act_dim, D)), (P + "action_pred.bias", (cfg.act_dim,))]
    else:                                                           # score_gpts.py:186-190
        out += [(P + "action_pred.0.weight", (100, D)), (P + "action_pred.0.bias", (100,)),
                (P + "action_pred.2.weight", (cfg.act_dim, 100)), (P + "action_pred.2.bias", (cfg.act_dim,))]
    return out


def n_params(cfg: ScoreGPTConfig) -> int:
    return int(sum(int(np.prod(s)) for _, s in param_shapes(cfg)))


def make_weights(cfg: ScoreGPTConfig, seed: int = 0, std: float = 0.02,
                 bias_std: Optional[float] = None) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights (the trained checkpoints are not shipped: .MISSING_LARGE_BLOBS).

    Follows the *distribution* of the reference init (score_gpts.py:202-211: Linear/pos_emb
    N(0, std), LN weight 1) but, so that every term of the forward is exercised by the parity
    tests, biases and LN affine parameters are perturbed too (bias_std defaults to std).
    One numpy PCG64 stream per tensor, seeded with (seed, index): identical on every machine.
    """
    bias_std = std if bias_std is None else bias_std
    w = {}
    for idx, (name, shape) in enumerate(param_shapes(cfg)):
        rng = np.random.Generator(np.random.PCG64([seed, idx]))
        x = rng.standard_normal(shape, dtype=np.float32)
        if ".ln" in name or "ln_f" in name:
            x = (1.0 + 0.1 * x) if name.endswith("weight") else 0.1 * x
        elif name.endswith("bias"):
            x = bias_std * x
        else:
            x = std * x
        w[name] = np.ascontiguousarray(x, dtype=np.float32)
    return w


# --------------------------------------------------------------------------------------------
# elementary ops (torch semantics)
# --------------------------------------------------------------------------------------------
def _linear(x, w, b):
    return x @ w.T + b                                      # nn.Linear


def _layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(axis=-1, keepdims=True)                     # nn.LayerNorm: biased variance, eps 1e-5
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(eps)) * w + b


def _gelu_erf(x):
    # nn.GELU() default = exact erf form (score_gpts.py:107)
    return (x.dtype.type(0.5) * x * (x.dtype.type(1.0) + _erf(x * x.dtype.type(1.0 / math.sqrt(2.0))))).astype(x.dtype)


def _silu(x):
    return x / (x.dtype.type(1.0) + np.exp(-x))


def _softmax_lastdim(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


# --------------------------------------------------------------------------------------------
# DiffusionGPT.forward  (score_gpts.py:272-358), eval mode (dropout off, no mask_cond)
# --------------------------------------------------------------------------------------------
def score_gpt_forward(w: Dict[str, np.ndarray], cfg: ScoreGPTConfig, states, actions, goals, sigma,
                      uncond: bool = False, dtype=np.float32, return_hidden: bool = False):
    P = "inner_model."
    W = {k: v.astype(dtype) for k, v in w.items()} if dtype != np.float32 else w
    states = np.asarray(states, dtype=dtype)
    actions = np.asarray(actions, dtype=dtype)
    sigma = np.asarray(sigma, dtype=dtype).reshape(-1)
    b, t, _ = states.shape
    D, H, G = cfg.embed_dim, cfg.n_heads, cfg.G
    assert t <= cfg.block_size                                           # :282
    # sigma embedding: Linear(1, D) of log(sigma)/4                      # :284-288
    emb_t = _linear((np.log(sigma) / dtype(4)).reshape(b, 1), W[P + "sigma_emb.weight"], W[P + "sigma_emb.bias"])
    emb_t = emb_t.reshape(b, 1, D)
    pos = W[P + "pos_emb"][:, : t + G, :]                                # :311-318
    state_x = _linear(states, W[P + "tok_emb.weight"], W[P + "tok_emb.bias"]) + pos[:, G:, :]     # :305,323
    action_x = _linear(actions, W[P + "action_emb.weight"], W[P + "action_emb.bias"]) + pos[:, G:, :]  # :307,325
    sa = np.stack([state_x, action_x], axis=2).reshape(b, 2 * t, D)      # :330-331 -> [s1,a1,s2,a2,..]
    if cfg.goal_conditioned:
        goals = np.asarray(goals, dtype=dtype)
        if goals.ndim == 2:
            goals = np.broadcast_to(goals[None], (b,) + goals.shape)
        if uncond:
            goals = np.zeros_like(goals)                                 # :301-302
        goal_x = _linear(goals, W[P + "tok_emb.weight"], W[P + "tok_emb.bias"]) + pos[:, :G, :]   # :306,322
        if goal_x.shape[0] != b:
            goal_x = np.broadcast_to(goal_x, (b,) + goal_x.shape[1:])
        x = np.concatenate([emb_t, goal_x, sa], axis=1)                  # :335
    else:
        x = np.concatenate([emb_t, sa], axis=1)                          # :337
    T = x.shape[1]
    hd = D // H
    causal = np.tril(np.ones((T, T), dtype=bool))                        # :42-47
    for i in range(cfg.n_layers):                                        # Block.forward :112-115
        B_ = f"{P}blocks.{i}."
        h = _layer_norm(x, W[B_ + "ln1.weight"], W[B_ + "ln1.bias"])
        k = _linear(h, W[B_ + "attn.key.weight"], W[B_ + "attn.key.bias"]).reshape(b, T, H, hd).transpose(0, 2, 1, 3)
        q = _linear(h, W[B_ + "attn.query.weight"], W[B_ + "attn.query.bias"]).reshape(b, T, H, hd).transpose(0, 2, 1, 3)
        v = _linear(h, W[B_ + "attn.value.weight"], W[B_ + "attn.value.bias"]).reshape(b, T, H, hd).transpose(0, 2, 1, 3)
        att = (q @ k.transpose(0, 1, 3, 2)) * dtype(1.0 / math.sqrt(hd))  # :69
        att = np.where(causal, att, dtype(-np.inf))                      # :70
        att = _softmax_lastdim(att)                                      # :71
        y = (att @ v).transpose(0, 2, 1, 3).reshape(b, T, D)             # :73-76
        x = x + _linear(y, W[B_ + "attn.proj.weight"], W[B_ + "attn.proj.bias"])   # :79,113
        h = _layer_norm(x, W[B_ + "ln2.weight"], W[B_ + "ln2.bias"])
        h = _gelu_erf(_linear(h, W[B_ + "mlp.0.weight"], W[B_ + "mlp.0.bias"]))
        x = x + _linear(h, W[B_ + "mlp.2.weight"], W[B_ + "mlp.2.bias"])            # :114
    x = _layer_norm(x, W[P + "ln_f.weight"], W[P + "ln_f.bias"])         # :341
    hidden = x
    x = x[:, (G + 1):, :]                                                # :344 (second_half_idx)
    x = x.reshape(b, x.shape[1] // 2, 2, D)                              # :347-351
    a_out = x[:, :, 1, :]                                                # :353 action tokens
    if cfg.linear_output:
        pred = _linear(a_out, W[P + "action_pred.weight"], W[P + "action_pred.bias"])            # :354
    else:
        pred = _linear(_silu(_linear(a_out, W[P + "action_pred.0.weight"], W[P + "action_pred.0.bias"])),
                       W[P + "action_pred.2.weight"], W[P + "action_pred.2.bias"])
    if return_hidden:
        return pred, hidden
    return pred


# --------------------------------------------------------------------------------------------
# GCDenoiser (score_wrappers.py:31-96)
# --------------------------------------------------------------------------------------------
def get_scalings(sigma, sigma_data):
    """c_skip, c_out, c_in  (score_wrappers.py:40-42)."""
    sd = sigma.dtype.type(sigma_data)
    c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
    c_out = sigma * sd / (sigma ** 2 + sd ** 2) ** sigma.dtype.type(0.5)
    c_in = 1 / (sigma ** 2 + sd ** 2) ** sigma.dtype.type(0.5)
    return c_skip, c_out, c_in


def denoise(w, cfg: ScoreGPTConfig, state, action, goal, sigma, uncond=False, dtype=np.float32):
    """GCDenoiser.forward (score_wrappers.py:95-96)."""
    action = np.asarray(action, dtype=dtype)
    sigma = np.asarray(sigma, dtype=dtype).reshape(-1)
    c_skip, c_out, c_in = [s.reshape(-1, 1, 1) for s in get_scalings(sigma, cfg.sigma_data)]
    f = score_gpt_forward(w, cfg, state, action * c_in, goal, sigma, uncond=uncond, dtype=dtype)
    return f * c_out + action * c_skip


def denoise_cfg(w, cfg: ScoreGPTConfig, state, action, goal, sigma, cond_lambda, dtype=np.float32):
    """ClassifierFreeSampleModel.forward (classifier_free_sampler.py:35-49)."""
    if cond_lambda == 1:
        return denoise(w, cfg, state, action, goal, sigma, dtype=dtype)
    if cond_lambda == 0:
        return denoise(w, cfg, state, action, goal, sigma, uncond=True, dtype=dtype)
    out = denoise(w, cfg, state, action, goal, sigma, dtype=dtype)
    out_u = denoise(w, cfg, state, action, goal, sigma, uncond=True, dtype=dtype)
    return out_u + dtype(cond_lambda) * (out - out_u)


def score_matching_loss(w, cfg: ScoreGPTConfig, state, action, goal, noise, sigma, dtype=np.float32):
    """GCDenoiser.loss with dropout/goal-masking off (score_wrappers.py:70-79)."""
    action = np.asarray(action, dtype=dtype)
    noise = np.asarray(noise, dtype=dtype)
    sigma = np.asarray(sigma, dtype=dtype).reshape(-1)
    noised = action + noise * sigma.reshape(-1, 1, 1)
    c_skip, c_out, c_in = [s.reshape(-1, 1, 1) for s in get_scalings(sigma, cfg.sigma_data)]
    out = score_gpt_forward(w, cfg, state, noised * c_in, goal, sigma, dtype=dtype)
    target = (action - c_skip * noised) / c_out
    return ((out - target) ** 2).reshape(action.shape[0], -1).mean()


def make_model(w, cfg: ScoreGPTConfig, cond_lambda: Optional[float] = None, dtype=np.float32) -> Callable:
    """A ``model(state, action, goal, sigma)`` callable as the samplers expect it."""
    if cond_lambda is None:
        return lambda s, a, g, sig, **kw: denoise(w, cfg, s, a, g, sig, dtype=dtype, **kw)
    return lambda s, a, g, sig, **kw: denoise_cfg(w, cfg, s, a, g, sig, cond_lambda, dtype=dtype)


# --------------------------------------------------------------------------------------------
# noise schedules (gc_sampling.py:22-95).  fp32 like torch; returns n+1 values, last = 0
# --------------------------------------------------------------------------------------------
def _linspace32(a, b, n):
    # torch.linspace(fp32) computes start + i*step for the first half and end - (n-1-i)*step for
    # the second half, in fp32.
    a32, b32 = np.float32(a), np.float32(b)
    if n == 1:
        return np.array([a32], dtype=np.float32)
    step = np.float32((b32 - a32) / np.float32(n - 1))
    i = np.arange(n)
    lo = (a32 + step * i.astype(np.float32)).astype(np.float32)
    hi = (b32 - step * (n - 1 - i).astype(np.float32)).astype(np.float32)
    return np.where(i < n // 2, lo, hi).astype(np.float32)


def _append_zero(x):
    return np.concatenate([x.astype(np.float32), np.zeros(1, np.float32)])


def get_sigmas_exponential(n, sigma_min, sigma_max):
    return _append_zero(np.exp(_linspace32(math.log(sigma_max), math.log(sigma_min), n)))      # :35-38


def get_sigmas_linear(n, sigma_min, sigma_max):
    return _append_zero(_linspace32(sigma_max, sigma_min, n))                                 # :41-44


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0):
    ramp = _linspace32(0, 1, n)                                                               # :26-32
    min_inv, max_inv = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return _append_zero((np.float32(max_inv) + ramp * np.float32(min_inv - max_inv)) ** np.float32(rho))


def get_sigmas_polyexponential(n, sigma_min, sigma_max, rho=1.0):
    ramp = _linspace32(1, 0, n) ** np.float32(rho)                                            # :91-95
    return _append_zero(np.exp(ramp * np.float32(math.log(sigma_max) - math.log(sigma_min))
                               + np.float32(math.log(sigma_min))))


def get_sigmas_vp(n, beta_d=19.9, beta_min=0.1, eps_s=1e-3):
    t = _linspace32(1, eps_s, n)                                                              # :84-88
    return _append_zero(np.sqrt(np.exp(np.float32(beta_d) * t ** 2 / 2 + np.float32(beta_min) * t) - 1))


def get_sigmas_ve(n, sigma_min=0.02, sigma_max=100):
    t = _linspace32(0, n + 1, n)                                                              # :61-68
    t = np.float32(sigma_max ** 2) * (np.float32(sigma_min ** 2 / sigma_max ** 2) ** (t / np.float32(n - 1)))
    return _append_zero(np.sqrt(t))


def cosine_beta_schedule(n, s=0.008):
    steps = n + 1                                                                             # :47-58
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = np.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
    return _append_zero(np.flip(betas).astype(np.float32))


SCHEDULES = {"exponential": get_sigmas_exponential, "linear": get_sigmas_linear,
             "karras": get_sigmas_karras}


# --------------------------------------------------------------------------------------------
# samplers (gc_sampling.py).  sigma arithmetic in fp32 scalars like the reference's 0-d tensors.
# --------------------------------------------------------------------------------------------
def _ones(action):
    return np.ones(action.shape[0], dtype=action.dtype)


def sample_ddim(model, state, action, goal, sigmas):
    """gc_sampling.py:895-924."""
    sigmas = np.asarray(sigmas, dtype=np.float32)
    action = np.asarray(action)
    dt_ = action.dtype.type
    with np.errstate(divide="ignore"):
        for i in range(len(sigmas) - 1):
            denoised = model(state, action, goal, sigmas[i] * _ones(action))
            t, t_next = -np.log(sigmas[i]), -np.log(sigmas[i + 1])                 # :921
            h = t_next - t
            action = dt_(np.exp(-t_next) / np.exp(-t)) * action - dt_(np.expm1(-h)) * denoised   # :923
    return action


def sample_euler(model, state, action, goal, sigmas, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"),
                 s_noise=1.0, eps_list=None):
    """gc_sampling.py:167-213.  ``eps_list`` injects the churn noise (RNG parity is by injection)."""
    sigmas = np.asarray(sigmas, dtype=np.float32)
    action = np.asarray(action)
    dt_ = action.dtype.type
    n = len(sigmas) - 1
    for i in range(n):
        gamma = min(s_churn / n, 2 ** 0.5 - 1) if s_tmin <= sigmas[i] <= s_tmax else 0.0
        sigma_hat = np.float32(sigmas[i] * np.float32(gamma + 1))
        if gamma > 0:
            action = action + eps_list[i] * dt_(s_noise) * dt_((sigma_hat ** 2 - sigmas[i] ** 2) ** 0.5)
        denoised = model(state, action, goal, sigma_hat * _ones(action))
        d = (action - denoised) / dt_(sigma_hat)                                   # to_d :98-100
        dt = sigmas[i + 1] - sigma_hat
        action = action + d * dt_(dt)
    return action


def sample_heun(model, state, action, goal, sigmas, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"),
                s_noise=1.0, eps_list=None):
    """gc_sampling.py:259-314."""
    sigmas = np.asarray(sigmas, dtype=np.float32)
    action = np.asarray(action)
    dt_ = action.dtype.type
    n = len(sigmas) - 1
    for i in range(n):
        gamma = min(s_churn / n, 2 ** 0.5 - 1) if s_tmin <= sigmas[i] <= s_tmax else 0.0
        sigma_hat = np.float32(sigmas[i] * np.float32(gamma + 1))
        if gamma > 0:
            action = action + eps_list[i] * dt_(s_noise) * dt_((sigma_hat ** 2 - sigmas[i] ** 2) ** 0.5)
        denoised = model(state, action, goal, sigma_hat * _ones(action))
        d = (action - denoised) / dt_(sigma_hat)
        dt = dt_(sigmas[i + 1] - sigma_hat)
        if sigmas[i + 1] == 0:
            action = action + d * dt                                               # :301-303
        else:
            action_2 = action + d * dt
            denoised_2 = model(state, action_2, goal, sigmas[i + 1] * _ones(action))
            d_2 = (action_2 - denoised_2) / dt_(sigmas[i + 1])
            action = action + (d + d_2) / dt_(2) * dt                              # :309-310
    return action


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    """gc_sampling.py:107-114 (fp32 scalars)."""
    if not eta:
        return sigma_to, np.float32(0.0)
    sigma_from, sigma_to = np.float32(sigma_from), np.float32(sigma_to)
    sigma_up = min(sigma_to, np.float32(eta) * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** np.float32(0.5))
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** np.float32(0.5)
    return np.float32(sigma_down), np.float32(sigma_up)


def sample_euler_ancestral(model, state, action, goal, sigmas, eta=1.0, noise_list=None):
    """gc_sampling.py:216-256; the per-step randn is injected through ``noise_list``."""
    sigmas = np.asarray(sigmas, dtype=np.float32)
    action = np.asarray(action)
    dt_ = action.dtype.type
    for i in range(len(sigmas) - 1):
        denoised = model(state, action, goal, sigmas[i] * _ones(action))
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1], eta=eta)
        d = (action - denoised) / dt_(sigmas[i])
        action = action + d * dt_(sigma_down - sigmas[i])
        if sigma_down > 0:
            action = action + noise_list[i] * dt_(sigma_up)
    return action


def sample_dpmpp_2m(model, state, action, goal, sigmas):
    """gc_sampling.py:702-736."""
    sigmas = np.asarray(sigmas, dtype=np.float32)
    action = np.asarray(action)
    dt_ = action.dtype.type
    old = None
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(len(sigmas) - 1):
            denoised = model(state, action, goal, sigmas[i] * _ones(action))
            t, t_next = -np.log(sigmas[i]), -np.log(sigmas[i + 1])
            h = t_next - t
            if old is None or sigmas[i + 1] == 0:
                action = dt_(np.exp(-t_next) / np.exp(-t)) * action - dt_(np.expm1(-h)) * denoised
            else:
                h_last = t - (-np.log(sigmas[i - 1]))
                r = h_last / h
                dd = dt_(1 + 1 / (2 * r)) * denoised - dt_(1 / (2 * r)) * old
                action = dt_(np.exp(-t_next) / np.exp(-t)) * action - dt_(np.expm1(-h)) * dd
            old = denoised
    return action


def sample_dpm_2(model, state, action, goal, sigmas):
    """gc_sampling.py:317-375 with s_churn = 0."""
    sigmas = np.asarray(sigmas, dtype=np.float32)
    action = np.asarray(action)
    dt_ = action.dtype.type
    for i in range(len(sigmas) - 1):
        sigma_hat = sigmas[i]
        denoised = model(state, action, goal, sigma_hat * _ones(action))
        d = (action - denoised) / dt_(sigma_hat)
        if sigmas[i + 1] == 0:
            action = action + d * dt_(sigmas[i + 1] - sigma_hat)
        else:
            # torch.lerp(a, b, 0.5) = a + 0.5 * (b - a)
            la, lb = np.log(sigma_hat), np.log(sigmas[i + 1])
            sigma_mid = np.exp(np.float32(la + np.float32(0.5) * (lb - la)))
            dt_1 = sigma_mid - sigma_hat
            dt_2 = sigmas[i + 1] - sigma_hat
            action_2 = action + d * dt_(dt_1)
            denoised_2 = model(state, action_2, goal, sigma_mid * _ones(action))
            d_2 = (action_2 - denoised_2) / dt_(sigma_mid)
            action = action + d_2 * dt_(dt_2)
    return action


def sample_dpmpp_2s(model, state, action, goal, sigmas):
    """gc_sampling.py:928-966."""
    sigmas = np.asarray(sigmas, dtype=np.float32)
    action = np.asarray(action)
    dt_ = action.dtype.type
    for i in range(len(sigmas) - 1):
        denoised = model(state, action, goal, sigmas[i] * _ones(action))
        if sigmas[i + 1] == 0:
            d = (action - denoised) / dt_(sigmas[i])
            action = action + d * dt_(sigmas[i + 1] - sigmas[i])
        else:
            t, t_next = -np.log(sigmas[i]), -np.log(sigmas[i + 1])
            r = np.float32(0.5)
            h = t_next - t
            s = t + r * h
            x_2 = dt_(np.exp(-s) / np.exp(-t)) * action - dt_(np.expm1(-h * r)) * denoised
            denoised_2 = model(state, x_2, goal, np.exp(-s) * _ones(action))
            action = dt_(np.exp(-t_next) / np.exp(-t)) * action - dt_(np.expm1(-h)) * denoised_2
    return action


SAMPLERS = {"ddim": sample_ddim, "euler": sample_euler, "heun": sample_heun,
            "euler_ancestral": sample_euler_ancestral, "dpmpp_2m": sample_dpmpp_2m,
            "dpm": sample_dpm_2, "dpmpp_2s": sample_dpmpp_2s}


# --------------------------------------------------------------------------------------------
# training feed: TrajectorySlicerDataset (envs/dataloaders/trajectory_loader.py:77-197)
# --------------------------------------------------------------------------------------------
def window_table(lengths, window):
    """The slicer's (trajectory, start, end) rows (:126-135)."""
    rows = []
    for i, n in enumerate(lengths):
        if int(n) - window >= 0:
            rows += [(i, start, start + window) for start in range(int(n) - window + 1)]
    return np.asarray(rows, dtype=np.int32).reshape(-1, 3)


def slice_windows(observations, actions, lengths, slices, ids, goal_len=0, min_future_sep=0, mode="random", draws=None):
    """``[dataset[i] for i in ids]`` collated (:160-197).  ``draws[k]`` is the value np.random.randint(lo, hi) - lo
    took for item k (the injected randomness of the random future-goal mode)."""
    t_max = observations.shape[1]
    obs_out, act_out, goal_out = [], [], []
    for k, idx in enumerate(ids):
        i, start, end = (int(v) for v in slices[int(idx)])
        obs_out.append(observations[i, start:end])
        act_out.append(actions[i, start:end])
        if goal_len > 0:
            lo, hi = end + min_future_sep, int(lengths[i]) - goal_len
            if lo < hi:
                if mode == "tail":
                    g = observations[i, t_max - goal_len:]            # the PADDED tensor's tail (:176)
                elif mode == "seq_end":
                    g = observations[i, end:end + goal_len]
                else:
                    g0 = lo + int(draws[k]) % (hi - lo)
                    g = observations[i, g0:g0 + goal_len]
            else:
                g = np.zeros((goal_len, observations.shape[2]), dtype=observations.dtype)
            goal_out.append(g)
    out = {"observation": np.stack(obs_out), "action": np.stack(act_out)}
    if goal_len > 0:
        out["goal_observation"] = np.stack(goal_out)
    return out


# --------------------------------------------------------------------------------------------
# training-side sigma density (utils.py:173-185)
# --------------------------------------------------------------------------------------------
def log_logistic_from_uniform(u, loc, scale, min_value, max_value):
    """rand_log_logistic (utils.py:178-185) with the uniform draw ``u`` (float64 in [0,1)) injected."""
    u = np.asarray(u, dtype=np.float64)
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    min_cdf = sig((math.log(min_value) - loc) / scale)
    max_cdf = sig((math.log(max_value) - loc) / scale)
    u = u * (max_cdf - min_cdf) + min_cdf
    return np.exp(np.log(u / (1 - u)) * scale + loc).astype(np.float32)


# --------------------------------------------------------------------------------------------
# synthetic inputs -- shared by tests and bench so both sides see the same data
# --------------------------------------------------------------------------------------------
def make_inputs(cfg: ScoreGPTConfig, batch: int, seed: int = 0, t: Optional[int] = None,
                sigma_max: float = 1.0):
    t = cfg.obs_seq_len if t is None else t
    rng = np.random.Generator(np.random.PCG64([seed, 9001]))
    state = rng.standard_normal((batch, t, cfg.obs_dim), dtype=np.float32)
    goal = rng.standard_normal((batch, max(cfg.goal_seq_len, 1), cfg.obs_dim), dtype=np.float32)
    x_t = rng.standard_normal((batch, t, cfg.act_dim), dtype=np.float32) * np.float32(sigma_max)
    return state, goal, x_t
