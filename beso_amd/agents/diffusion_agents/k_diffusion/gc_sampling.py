"""Drop-in for ``beso...k_diffusion.gc_sampling``: noise schedules and the iterative samplers of the
goal-conditioned score model (reference: gc_sampling.py).

Signature of every sampler, as in the reference:
    sample_X(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None,
             disable=None, ...) -> action
where ``model(state, action, goal, sigma[B], **extra_args) -> [B, t, act]``.

MI355X-first differences that do not change results:
  * the sigma schedule is read back to the HOST once and all step coefficients are computed there
    as fp32 scalars, so the loop never branches on a device tensor (the reference syncs on every
    ``sigmas[i + 1] == 0`` / ``s_tmin <= sigmas[i]`` test: gc_sampling.py:198,289,301,360);
  * when ``model`` is a ``beso_amd`` GCDenoiser (optionally inside ClassifierFreeSampleModel) and the
    call is the plain one (no churn, no callback, no scaler, no extra args),
    ddim / euler / heun / euler_ancestral run as ONE enqueue of the whole loop through ``beso_sample`` / ``beso_sample_ancestral``
    (include/beso_hip.h) -- otherwise the generic loops below call ``model`` once per evaluation.
"""
import math

import numpy as np
import torch
from scipy import integrate

from . import utils
from .classifier_free_sampler import ClassifierFreeSampleModel
from .score_wrappers import GCDenoiser

f32 = np.float32


# ------------------------------------------------------------------------------------------------
# noise schedules (gc_sampling.py:22-95): n values + a trailing zero, fp32
# ------------------------------------------------------------------------------------------------
def append_zero(action):
    return torch.cat([action, action.new_zeros([1])])


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7., device='cpu'):
    """Karras et al. (2022) schedule (:26-32)."""
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return append_zero((hi + torch.linspace(0, 1, n) * (lo - hi)) ** rho).to(device)


def get_sigmas_exponential(n, sigma_min, sigma_max, device='cpu'):
    """Log-linear schedule (:35-38)."""
    return append_zero(torch.linspace(math.log(sigma_max), math.log(sigma_min), n, device=device).exp())


def get_sigmas_linear(n, sigma_min, sigma_max, device='cpu'):
    """Linear schedule (:41-44)."""
    return append_zero(torch.linspace(sigma_max, sigma_min, n, device=device))


def cosine_beta_schedule(n, s=0.008, device='cpu'):
    """Cosine beta schedule (:47-58)."""
    steps = n + 1
    grid = np.linspace(0, steps, steps)
    abar = np.cos(((grid / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    abar = abar / abar[0]
    betas = np.clip(1 - abar[1:] / abar[:-1], a_min=0, a_max=0.999)
    return append_zero(torch.tensor(np.flip(betas).copy(), device=device, dtype=torch.float32))


def get_sigmas_ve(n, sigma_min=0.02, sigma_max=100, device='cpu'):
    """(:61-68)"""
    t = torch.linspace(0, n + 1, n, device=device)
    return append_zero(torch.sqrt((sigma_max ** 2) * ((sigma_min ** 2 / sigma_max ** 2) ** (t / (n - 1)))))


def get_iddpm_sigmas(n, sigma_min=0.02, sigma_max=100, M=1000, j_0=0, C_1=0.001, C_2=0.008, device='cpu'):
    """(:71-81)"""
    idx = torch.arange(n, dtype=torch.float64, device=device)
    u = torch.zeros(M + 1, dtype=torch.float64, device=device)
    abar = lambda j: (0.5 * np.pi * j / M / (C_2 + 1)).sin() ** 2        # noqa: E731
    for j in torch.arange(M, j_0, -1, device=device):
        u[j - 1] = ((u[j] ** 2 + 1) / (abar(j - 1) / abar(j)).clip(min=C_1) - 1).sqrt()
    u = u[torch.logical_and(u >= sigma_min, u <= sigma_max)]
    return append_zero(u[((len(u) - 1) / (n - 1) * idx).round().to(torch.int64)]).to(torch.float32)


def get_sigmas_vp(n, beta_d=19.9, beta_min=0.1, eps_s=1e-3, device='cpu'):
    """(:84-88)"""
    t = torch.linspace(1, eps_s, n, device=device)
    return append_zero(torch.sqrt(torch.exp(beta_d * t ** 2 / 2 + beta_min * t) - 1))


def get_sigmas_polyexponential(n, sigma_min, sigma_max, rho=1., device='cpu'):
    """(:91-95)"""
    ramp = torch.linspace(1, 0, n, device=device) ** rho
    return append_zero(torch.exp(ramp * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min)))


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def to_d(action, sigma, denoised):
    """Karras ODE derivative (x - D(x)) / sigma (:98-100)."""
    if not torch.is_tensor(sigma):
        return (action - denoised) / float(sigma)
    return (action - denoised) / utils.append_dims(sigma, action.ndim)


def default_noise_sampler(x):
    return lambda sigma, sigma_next: torch.randn_like(x)


def get_ancestral_step(sigma_from, sigma_to, eta=1.):
    """sigma_down / sigma_up of an ancestral step (:107-114), fp32 scalars."""
    if not eta:
        return sigma_to, 0.
    sf, st = f32(sigma_from), f32(sigma_to)
    up = min(st, f32(eta) * (st ** 2 * (sf ** 2 - st ** 2) / sf ** 2) ** f32(0.5))
    down = (st ** 2 - up ** 2) ** f32(0.5)
    return f32(down), f32(up)


def _host_sigmas(sigmas):
    """The schedule as host fp32 scalars: the single device->host read of a sampling loop."""
    if torch.is_tensor(sigmas):
        return sigmas.detach().to('cpu', torch.float32).numpy()
    return np.asarray(sigmas, dtype=np.float32)


def _sig_vec(action, value):
    return action.new_full([action.shape[0]], float(value))


def _neg_log(s):
    with np.errstate(divide='ignore'):
        return -np.log(f32(s))


def _fused_target(model):
    """(GCDenoiser, cond_lambda) when ``model`` is one the HIP sampler loop understands."""
    if isinstance(model, GCDenoiser):
        return model, 1.0
    if isinstance(model, ClassifierFreeSampleModel) and isinstance(model.model, GCDenoiser):
        return model.model, float(model.cond_lambda)
    return None, None


def _try_fused(name, model, state, action, goal, sigmas, scaler, extra_args, callback):
    if scaler is not None or callback is not None or extra_args:
        return None
    den, lam = _fused_target(model)
    if den is None or not action.is_cuda:
        return None
    sig = _host_sigmas(sigmas)
    if not _interior_positive(sig):
        return None                    # e.g. get_sigmas_linear(sigma_min=0), the clipped cosine_beta: the stepwise loop serves them
    return den.fused_sampler(name, state, action, goal, sig, cond_lambda=lam)


def _interior_positive(sig) -> bool:
    """beso_sample / beso_sample_ancestral take schedules whose values are positive up to the trailing one."""
    return len(sig) >= 2 and all(float(v) > 0.0 for v in sig[:-1])


def _churn(i, n, sig, s_churn, s_tmin, s_tmax):
    gamma = min(s_churn / n, 2 ** 0.5 - 1) if s_tmin <= sig[i] <= s_tmax else 0.
    return gamma, f32(sig[i] * f32(gamma + 1))


# ------------------------------------------------------------------------------------------------
# samplers
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def sample_euler(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None,
                 s_churn=0., s_tmin=0., s_tmax=float('inf'), s_noise=1.):
    """Algorithm 2 of Karras et al. (2022) without the second-order correction (:167-213).
    The generic loop draws ``eps`` every step like the reference (:199), used or not; the fused
    loop draws nothing (with s_churn = 0 the draws never enter the result)."""
    extra_args = {} if extra_args is None else extra_args
    if not s_churn:
        fused = _try_fused('euler', model, state, action, goal, sigmas, scaler, extra_args, callback)
        if fused is not None:
            return fused
    sig = _host_sigmas(sigmas)
    n = len(sig) - 1
    for i in range(n):
        gamma, sigma_hat = _churn(i, n, sig, s_churn, s_tmin, s_tmax)
        eps = torch.randn_like(action) * s_noise
        if gamma > 0:
            action = action + eps * float((sigma_hat ** 2 - sig[i] ** 2) ** 0.5)
        denoised = model(state, action, goal, _sig_vec(action, sigma_hat), **extra_args)
        d = to_d(action, sigma_hat, denoised)
        if callback is not None:
            callback({'x': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sigma_hat, 'denoised': denoised})
        action = action + d * float(sig[i + 1] - sigma_hat)
        if scaler is not None:
            action = scaler.clip_output(action)
    return action


@torch.no_grad()
def sample_euler_ancestral(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None,
                           disable=None, eta=1.):
    """Euler steps to sigma_down, then fresh noise of scale sigma_up (:216-256).  The plain call on a ``beso_amd``
    denoiser runs as one enqueue (``beso_sample_ancestral``) with the same sequence of ``randn_like`` draws."""
    extra_args = {} if extra_args is None else extra_args
    if scaler is None and callback is None and not extra_args and eta >= 0:
        den, lam = _fused_target(model)
        if den is not None and action.is_cuda and _interior_positive(_host_sigmas(sigmas)):
            fused = den.fused_sampler('euler_ancestral', state, action, goal, _host_sigmas(sigmas), cond_lambda=lam, eta=eta)
            if fused is not None:
                return fused
    sig = _host_sigmas(sigmas)
    for i in range(len(sig) - 1):
        denoised = model(state, action, goal, _sig_vec(action, sig[i]), **extra_args)
        sigma_down, sigma_up = get_ancestral_step(sig[i], sig[i + 1], eta=eta)
        if callback is not None:
            callback({'x': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
        d = to_d(action, sig[i], denoised)
        action = action + d * float(sigma_down - sig[i])
        if sigma_down > 0:
            action = action + torch.randn_like(action) * float(sigma_up)
        if scaler is not None:
            action = scaler.clip_output(action)
    return action


@torch.no_grad()
def sample_heun(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None,
                s_churn=0., s_tmin=0., s_tmax=float('inf'), s_noise=1.):
    """Algorithm 2 of Karras et al. (2022): Euler predictor + trapezoid corrector; the last step
    (sigma_next == 0) is plain Euler (:259-314)."""
    extra_args = {} if extra_args is None else extra_args
    if not s_churn:
        fused = _try_fused('heun', model, state, action, goal, sigmas, scaler, extra_args, callback)
        if fused is not None:
            return fused
    sig = _host_sigmas(sigmas)
    n = len(sig) - 1
    for i in range(n):
        gamma, sigma_hat = _churn(i, n, sig, s_churn, s_tmin, s_tmax)
        eps = torch.randn_like(action) * s_noise
        if gamma > 0:
            action = action + eps * float((sigma_hat ** 2 - sig[i] ** 2) ** 0.5)
        denoised = model(state, action, goal, _sig_vec(action, sigma_hat), **extra_args)
        d = to_d(action, sigma_hat, denoised)
        if callback is not None:
            callback({'x': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sigma_hat, 'denoised': denoised})
        dt = float(sig[i + 1] - sigma_hat)
        if sig[i + 1] == 0:
            action = action + d * dt
        else:
            action_2 = action + d * dt
            denoised_2 = model(state, action_2, goal, _sig_vec(action, sig[i + 1]), **extra_args)
            d_2 = to_d(action_2, sig[i + 1], denoised_2)
            action = action + (d + d_2) / 2 * dt
        if scaler is not None:
            action = scaler.clip_output(action)
    return action


def _log_midpoint(a, b):
    la, lb = np.log(f32(a)), np.log(f32(b))
    return f32(np.exp(f32(la + f32(0.5) * (lb - la))))        # torch.lerp(a, b, 0.5) = a + 0.5 (b - a)


@torch.no_grad()
def sample_dpm_2(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None,
                 s_churn=0., s_tmin=0., s_tmax=float('inf'), s_noise=1.):
    """DPM-Solver-2-like midpoint steps in log sigma (:317-375)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host_sigmas(sigmas)
    n = len(sig) - 1
    for i in range(n):
        gamma, sigma_hat = _churn(i, n, sig, s_churn, s_tmin, s_tmax)
        eps = torch.randn_like(action) * s_noise
        if gamma > 0:
            action = action + eps * float((sigma_hat ** 2 - sig[i] ** 2) ** 0.5)
        denoised = model(state, action, goal, _sig_vec(action, sigma_hat), **extra_args)
        d = to_d(action, sigma_hat, denoised)
        if callback is not None:
            callback({'action': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sigma_hat, 'denoised': denoised})
        if sig[i + 1] == 0:
            action = action + d * float(sig[i + 1] - sigma_hat)
        else:
            sigma_mid = _log_midpoint(sigma_hat, sig[i + 1])
            action_2 = action + d * float(sigma_mid - sigma_hat)
            denoised_2 = model(state, action_2, goal, _sig_vec(action, sigma_mid), **extra_args)
            d_2 = to_d(action_2, sigma_mid, denoised_2)
            action = action + d_2 * float(sig[i + 1] - sigma_hat)
        if scaler is not None:
            action = scaler.clip_output(action)
    return action


@torch.no_grad()
def sample_dpm_2_ancestral(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None,
                           disable=None, eta=1.):
    """Ancestral variant of the DPM-Solver-2-like sampler (:378-413)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host_sigmas(sigmas)
    for i in range(len(sig) - 1):
        denoised = model(state, action, goal, _sig_vec(action, sig[i]), **extra_args)
        sigma_down, sigma_up = get_ancestral_step(sig[i], sig[i + 1], eta=eta)
        if callback is not None:
            callback({'x': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
        d = to_d(action, sig[i], denoised)
        if sigma_down == 0:
            action = action + d * float(sigma_down - sig[i])
        else:
            sigma_mid = _log_midpoint(sig[i], sigma_down)
            action_2 = action + d * float(sigma_mid - sig[i])
            denoised_2 = model(state, action_2, goal, _sig_vec(action, sigma_mid), **extra_args)
            d_2 = to_d(action_2, sigma_mid, denoised_2)
            action = action + d_2 * float(sigma_down - sig[i])
            action = action + torch.randn_like(action) * float(sigma_up)
        if scaler is not None:
            action = scaler.clip_output(action)
    return action


def linear_multistep_coeff(order, t, i, j):
    """Adams-Bashforth-style coefficient of the j-th stored derivative at step i (:416-429)."""
    if order - 1 > i:
        raise ValueError(f'Order {order} too high for step {i}')

    def basis(tau):
        prod = 1.
        for k in range(order):
            if k != j:
                prod *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return prod
    return integrate.quad(basis, t[i], t[i + 1], epsrel=1e-4)[0]


@torch.no_grad()
def sample_lms(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None,
               order=4):
    """Linear multistep sampler (:432-468)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host_sigmas(sigmas)
    history = []
    for i in range(len(sig) - 1):
        denoised = model(state, action, goal, _sig_vec(action, sig[i]), **extra_args)
        history.append(to_d(action, sig[i], denoised))
        if len(history) > order:
            history.pop(0)
        if callback is not None:
            callback({'x': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
        cur = min(i + 1, order)
        coeffs = [linear_multistep_coeff(cur, sig, i, j) for j in range(cur)]
        action = action + sum(c * d for c, d in zip(coeffs, reversed(history)))
        if scaler is not None:
            action = scaler.clip_output(action)
    return action


def _exp_step_coeffs(s_from, s_to):
    """(sigma_fn(t_to)/sigma_fn(t_from), expm1(-h)) of the exponential-integrator update
    x <- ratio * x - expm1(-h) * denoised, in fp32 like the reference's 0-d tensors (:921-923)."""
    t, t_next = _neg_log(s_from), _neg_log(s_to)
    h = t_next - t
    with np.errstate(over='ignore', invalid='ignore'):
        return float(f32(np.exp(-t_next)) / f32(np.exp(-t))), float(f32(np.expm1(-h))), t, t_next, h


@torch.no_grad()
def sample_dpmpp_2m(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None):
    """DPM-Solver++(2M) (:702-736)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host_sigmas(sigmas)
    old_denoised = None
    for i in range(len(sig) - 1):
        denoised = model(state, action, goal, _sig_vec(action, sig[i]), **extra_args)
        if callback is not None:
            callback({'action': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
        ratio, em1, t, t_next, h = _exp_step_coeffs(sig[i], sig[i + 1])
        if old_denoised is None or sig[i + 1] == 0:
            action = ratio * action - em1 * denoised
        else:
            r = (t - _neg_log(sig[i - 1])) / h
            blended = float(1 + 1 / (2 * r)) * denoised - float(1 / (2 * r)) * old_denoised
            action = ratio * action - em1 * blended
        old_denoised = denoised
    return action


@torch.no_grad()
def sample_ddim(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None,
                eta=1.):
    """DPM-Solver-1 / DDIM (:895-924).  The final step (sigma_next = 0) returns the denoised action."""
    extra_args = {} if extra_args is None else extra_args
    fused = _try_fused('ddim', model, state, action, goal, sigmas, None, extra_args, callback)
    if fused is not None:
        return fused
    sig = _host_sigmas(sigmas)
    for i in range(len(sig) - 1):
        denoised = model(state, action, goal, _sig_vec(action, sig[i]), **extra_args)
        if callback is not None:
            callback({'action': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
        ratio, em1, *_ = _exp_step_coeffs(sig[i], sig[i + 1])
        action = ratio * action - em1 * denoised
    return action


@torch.no_grad()
def sample_dpmpp_2s(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None,
                    eta=1.):
    """DPM-Solver++(2S) (:928-966)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host_sigmas(sigmas)
    for i in range(len(sig) - 1):
        denoised = model(state, action, goal, _sig_vec(action, sig[i]), **extra_args)
        if callback is not None:
            callback({'action': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
        if sig[i + 1] == 0:
            action = action + to_d(action, sig[i], denoised) * float(sig[i + 1] - sig[i])
        else:
            action = _dpmpp_2s_update(model, state, action, goal, denoised, sig[i], sig[i + 1], extra_args)
        if scaler is not None:
            action = scaler.clip_output(action)
    return action


def _dpmpp_2s_update(model, state, action, goal, denoised, s_from, s_to, extra_args):
    t, t_next = _neg_log(s_from), _neg_log(s_to)
    r = f32(0.5)
    h = t_next - t
    s = t + r * h
    x_2 = float(f32(np.exp(-s)) / f32(np.exp(-t))) * action - float(f32(np.expm1(-h * r))) * denoised
    denoised_2 = model(state, x_2, goal, _sig_vec(action, f32(np.exp(-s))), **extra_args)
    return float(f32(np.exp(-t_next)) / f32(np.exp(-t))) * action - float(f32(np.expm1(-h))) * denoised_2


@torch.no_grad()
def sample_dpmpp_2s_ancestral(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None,
                              disable=None, eta=1., s_noise=1., noise_sampler=None):
    """Ancestral DPM-Solver++(2S) (:969-1016)."""
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = default_noise_sampler(action) if noise_sampler is None else noise_sampler
    sig = _host_sigmas(sigmas)
    for i in range(len(sig) - 1):
        denoised = model(state, action, goal, _sig_vec(action, sig[i]), **extra_args)
        sigma_down, sigma_up = get_ancestral_step(sig[i], sig[i + 1], eta=eta)
        if callback is not None:
            callback({'action': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
        if sigma_down == 0:
            action = action + to_d(action, sig[i], denoised) * float(sigma_down - sig[i])
        else:
            action = _dpmpp_2s_update(model, state, action, goal, denoised, sig[i], sigma_down, extra_args)
        action = action + noise_sampler(sig[i], sig[i + 1]) * s_noise * float(sigma_up)
        if scaler is not None:
            action = scaler.clip_output(action)
    return action


# ------------------------------------------------------------------------------------------------
# Brownian-path noise for the SDE sampler (:117-165)
# ------------------------------------------------------------------------------------------------
class BrownianTreeNoiseSampler:
    """``sampler(sigma, sigma_next) -> (W(t1) - W(t0)) / sqrt|t1 - t0|`` with t = transform(sigma): increments of
    ONE Brownian path per sample, so that overlapping intervals are correlated the way the reference's
    torchsde-backed sampler makes them (:144-165).  torchsde is not a dependency here: the path is built lazily on
    the action's device -- W is drawn at every time that is asked for, conditioned on the values already drawn at
    its neighbours (Brownian bridge inside the known range, a free increment outside it).  Same law as a Brownian
    tree, different random numbers; ``seed`` (an int, or one int per batch row) makes it repeatable."""

    def __init__(self, x, sigma_min, sigma_max, seed=None, transform=lambda x: x):
        self.transform = transform
        self._like = x
        t0, t1 = float(transform(torch.as_tensor(sigma_min))), float(transform(torch.as_tensor(sigma_max)))
        if seed is None:
            seed = int(torch.randint(0, 2 ** 63 - 1, []).item())
        try:
            seeds = [int(v) for v in seed]
            if len(seeds) != x.shape[0]:
                raise ValueError("one seed per batch row expected")
        except TypeError:
            seeds = None
        self._gens = [torch.Generator(device=x.device).manual_seed(v) for v in (seeds if seeds is not None else [int(seed)])]
        self._batched = seeds is not None
        self._times = [min(t0, t1)]                       # sorted; W(first time) = 0
        self._w = [torch.zeros_like(x)]
        self._at(max(t0, t1))

    def _randn(self):
        x = self._like
        if not self._batched:
            return torch.randn(x.shape, generator=self._gens[0], device=x.device, dtype=x.dtype)
        return torch.stack([torch.randn(x.shape[1:], generator=g, device=x.device, dtype=x.dtype) for g in self._gens])

    def _at(self, t):
        import bisect
        k = bisect.bisect_left(self._times, t)
        if k < len(self._times) and self._times[k] == t:
            return self._w[k]
        if k == 0:                                         # before the first known time
            w = self._w[0] - self._randn() * math.sqrt(self._times[0] - t)
        elif k == len(self._times):                        # beyond the last
            w = self._w[-1] + self._randn() * math.sqrt(t - self._times[-1])
        else:                                              # bridge between the neighbours
            a, b = self._times[k - 1], self._times[k]
            f = (t - a) / (b - a)
            w = torch.lerp(self._w[k - 1], self._w[k], f) + self._randn() * math.sqrt((t - a) * (b - t) / (b - a))
        self._times.insert(k, t)
        self._w.insert(k, w)
        return w

    def __call__(self, sigma, sigma_next):
        t0, t1 = float(self.transform(torch.as_tensor(sigma))), float(self.transform(torch.as_tensor(sigma_next)))
        lo, hi = min(t0, t1), max(t0, t1)
        w = self._at(hi) - self._at(lo)
        return (w if t1 >= t0 else -w) / math.sqrt(abs(t1 - t0))


# ------------------------------------------------------------------------------------------------
# DPM-Solver (:498-672): eps-prediction steps of order 1/2/3 in t = -log(sigma), fixed and adaptive step size
# ------------------------------------------------------------------------------------------------
class PIDStepSizeController:
    """Step-size controller of the adaptive solver (:498-524): h <- h * limiter(prod_k (1/err_k)^b_k)."""

    def __init__(self, h, pcoeff, icoeff, dcoeff, order=1, accept_safety=0.81, eps=1e-8):
        self.h = h
        self.b = ((pcoeff + icoeff + dcoeff) / order, -(pcoeff + 2 * dcoeff) / order, dcoeff / order)
        self.accept_safety = accept_safety
        self.eps = eps
        self.errs = []

    @staticmethod
    def limiter(x):
        return 1 + math.atan(x - 1)

    def propose_step(self, error):
        inv = 1 / (float(error) + self.eps)
        self.errs = [inv, inv, inv] if not self.errs else [inv] + self.errs[1:]
        factor = self.limiter(math.prod(e ** b for e, b in zip(self.errs, self.b)))
        accept = factor >= self.accept_safety
        if accept:
            self.errs = [self.errs[0], self.errs[0], self.errs[1]]
        self.h *= factor
        return accept


def _sigma_of(t):
    return f32(np.exp(-f32(t)))


class DPMSolver:
    """DPM-Solver steps (arXiv 2206.00927; reference :527-672).  All times are host fp32 scalars, the network is
    evaluated through ``model`` (the HIP denoiser on a GPU), the state update is a handful of axpy's on the device."""

    def __init__(self, model, extra_args=None, eps_callback=None, info_callback=None):
        self.model = model
        self.extra_args = {} if extra_args is None else extra_args
        self.eps_callback = eps_callback
        self.info_callback = info_callback

    t = staticmethod(lambda sigma: _neg_log(sigma))
    sigma = staticmethod(_sigma_of)

    def eps(self, state, action, goal, t):
        sig = _sigma_of(t)
        out = (action - self.model(state, action, goal, _sig_vec(action, sig), **self.extra_args)) / float(sig)
        if self.eps_callback is not None:
            self.eps_callback()
        return out

    def steps(self, orders, state, action, goal, t, t_next, eps, r1=None):
        """The updates of every order in ``orders`` from (action, t) to t_next, sharing the network evaluations
        (the reference's eps_cache): eps at t is given; order 2 needs eps at t + r1 h, order 3 also at t + 2/3 h."""
        h = f32(t_next - t)
        sn, em1 = float(_sigma_of(t_next)), float(f32(np.expm1(h)))
        out = {}
        if 1 in orders:
            out[1] = action - (sn * em1) * eps
        if 2 in orders or 3 in orders:
            r1 = f32(r1 if r1 is not None else (1 / 3 if 3 in orders else 1 / 2))
            s1 = f32(t + r1 * h)
            u1 = action - float(_sigma_of(s1) * f32(np.expm1(f32(r1 * h)))) * eps
            d1 = self.eps(state, u1, goal, s1) - eps
            if 2 in orders:
                out[2] = action - (sn * em1) * eps - float(f32(sn) / (2 * r1) * f32(em1)) * d1
            if 3 in orders:
                r2 = f32(2 / 3)
                s2 = f32(t + r2 * h)
                e2 = f32(np.expm1(f32(r2 * h)))
                u2 = action - float(_sigma_of(s2) * e2) * eps - float(_sigma_of(s2) * (r2 / r1) * (e2 / f32(r2 * h) - 1)) * d1
                d2 = self.eps(state, u2, goal, s2) - eps
                out[3] = action - (sn * em1) * eps - float(f32(sn) / r2 * (f32(em1) / h - 1)) * d2
        return out

    def _ancestral_target(self, t, t_next, t_end, eta):
        """(t_next_, sigma_up) of a step with added noise (:614-619)."""
        if not eta:
            return t_next, 0.
        sd, _ = get_ancestral_step(_sigma_of(t), _sigma_of(t_next), eta)
        t_down = min(f32(t_end), _neg_log(sd))
        return t_down, float((_sigma_of(t_next) ** 2 - _sigma_of(t_down) ** 2) ** f32(0.5))

    def dpm_solver_fast(self, state, action, goal, t_start, t_end, nfe, eta=0., s_noise=1., noise_sampler=None):
        # the reference draws from randn_like whatever sampler it is given (:604), also when the noise is scaled by 0
        noise_sampler = default_noise_sampler(action)
        t_start, t_end = f32(t_start), f32(t_end)
        if not t_end > t_start and eta:
            raise ValueError('eta must be 0 for reverse sampling')
        m = nfe // 3 + 1
        ts = torch.linspace(float(t_start), float(t_end), m + 1, dtype=torch.float32).numpy()
        orders = [3] * (m - 2) + [2, 1] if nfe % 3 == 0 else [3] * (m - 1) + [nfe % 3]
        for i, order in enumerate(orders):
            t, t_next = ts[i], ts[i + 1]
            t_next_, su = self._ancestral_target(t, t_next, t_end, eta)
            eps = self.eps(state, action, goal, t)
            if self.info_callback is not None:
                self.info_callback({'x': action, 'i': i, 't': t, 't_up': t, 'denoised': action - float(_sigma_of(t)) * eps})
            action = self.steps((order,), state, action, goal, t, t_next_, eps, r1=None if order == 3 else 0.5)[order]
            action = action + (su * s_noise) * noise_sampler(_sigma_of(t), _sigma_of(t_next))
        return action

    def dpm_solver_adaptive(self, state, action, goal, t_start, t_end, order=3, rtol=0.05, atol=0.0078, h_init=0.05,
                            pcoeff=0., icoeff=1., dcoeff=0., accept_safety=0.81, eta=0., s_noise=1.):
        """Embedded pairs 1/2 or 2/3 with a PID step size (:629-672).  The error norm runs over the WHOLE batch, so
        the accepted step sequence depends on which samples share a call: one device->host scalar per step."""
        noise_sampler = default_noise_sampler(action)
        if order not in {2, 3}:
            raise ValueError('order should be 2 or 3')
        t_start, t_end = f32(t_start), f32(t_end)
        forward = t_end > t_start
        if not forward and eta:
            raise ValueError('eta must be 0 for reverse sampling')
        pid = PIDStepSizeController(abs(h_init) * (1 if forward else -1), pcoeff, icoeff, dcoeff,
                                    1.5 if eta else order, accept_safety)
        info = {'steps': 0, 'nfe': 0, 'n_accept': 0, 'n_reject': 0}
        s, action_prev = t_start, action
        while (s < t_end - 1e-5) if forward else (s > t_end + 1e-5):
            t = min(t_end, f32(s + f32(pid.h))) if forward else max(t_end, f32(s + f32(pid.h)))
            t_, su = self._ancestral_target(s, t, t_end, eta)
            eps = self.eps(state, action, goal, s)
            cand = self.steps((order - 1, order), state, action, goal, s, t_, eps, r1=1 / 3 if order == 3 else 1 / 2)
            low, high = cand[order - 1], cand[order]
            delta = (rtol * torch.maximum(low.abs(), action_prev.abs())).clamp_min(atol)
            error = torch.linalg.norm((low - high) / delta) / action.numel() ** 0.5
            if pid.propose_step(error):
                action_prev = low
                action = high + (su * s_noise) * noise_sampler(_sigma_of(s), _sigma_of(t))
                s = t
                info['n_accept'] += 1
            else:
                info['n_reject'] += 1
            info['nfe'] += order
            info['steps'] += 1
            if self.info_callback is not None:
                self.info_callback({'x': action, 'i': info['steps'] - 1, 't': s, 't_up': s,
                                    'denoised': action - float(_sigma_of(s)) * eps, 'error': error, 'h': pid.h, **info})
        return action, info


def _solver(model, extra_args, callback):
    solver = DPMSolver(model, extra_args)
    if callback is not None:
        solver.info_callback = lambda info: callback({'sigma': _sigma_of(info['t']), 'sigma_hat': _sigma_of(info['t_up']), **info})
    return solver


@torch.no_grad()
def sample_dpm_fast(model, state, action, goal, sigma_min, sigma_max, n, scaler=None, extra_args=None, callback=None,
                    disable=None, eta=0., s_noise=1., noise_sampler=None):
    """DPM-Solver-fast: ``n`` network evaluations in steps of order 3, 3, ..., (2, 1 | n mod 3), uniform in
    log sigma from sigma_max to sigma_min (:675-699)."""
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError('sigma_min and sigma_max must not be 0')
    return _solver(model, extra_args, callback).dpm_solver_fast(
        state, action, goal, _neg_log(sigma_max), _neg_log(sigma_min), n, eta, s_noise, noise_sampler)


@torch.no_grad()
def sample_dpm_adaptive(model, state, action, goal, sigma_min, sigma_max, extra_args=None, callback=None, disable=None,
                        order=3, rtol=0.05, atol=0.0078, h_init=0.05, pcoeff=0., icoeff=1., dcoeff=0.,
                        accept_safety=0.81, eta=0., s_noise=1., return_info=False):
    """DPM-Solver-12 / -23 with adaptive step size (:855-892)."""
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError('sigma_min and sigma_max must not be 0')
    action, info = _solver(model, extra_args, callback).dpm_solver_adaptive(
        state, action, goal, _neg_log(sigma_max), _neg_log(sigma_min), order, rtol, atol, h_init, pcoeff, icoeff,
        dcoeff, accept_safety, eta, s_noise)
    return (action, info) if return_info else action


@torch.no_grad()
def sample_dpmpp_sde(model, state, action, goal, sigmas, extra_args=None, callback=None, disable=None, eta=1.,
                     s_noise=1., scaler=None, noise_sampler=None, r=1 / 2):
    """Stochastic DPM-Solver++ (:739-795): two exponential-integrator half steps per sigma interval, each to the
    ancestral sigma_down of its sub-interval followed by the matching amount of Brownian noise.  This is what
    ``sampler_type='dpmpp_2m_sde'`` runs (beso_agent.py:452-453)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host_sigmas(sigmas)
    if noise_sampler is None:
        noise_sampler = BrownianTreeNoiseSampler(action, sig[sig > 0].min(), sig.max())
    r = f32(r)
    fac = float(1 / (2 * r))

    def to(x, denoised, t, t_to):                          # x at t -> t_to along the denoised direction
        return float(_sigma_of(t_to) / _sigma_of(t)) * x - float(f32(np.expm1(f32(t - t_to)))) * denoised

    for i in range(len(sig) - 1):
        denoised = model(state, action, goal, _sig_vec(action, sig[i]), **extra_args)
        if callback is not None:
            callback({'x': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
        if sig[i + 1] == 0:
            action = action + to_d(action, sig[i], denoised) * float(sig[i + 1] - sig[i])
            continue
        t, t_next = _neg_log(sig[i]), _neg_log(sig[i + 1])
        s = f32(t + f32(t_next - t) * r)
        sd, su = get_ancestral_step(_sigma_of(t), _sigma_of(s), eta)
        x_2 = to(action, denoised, t, _neg_log(sd))
        x_2 = x_2 + noise_sampler(_sigma_of(t), _sigma_of(s)) * (s_noise * float(su))
        denoised_2 = model(state, x_2, goal, _sig_vec(action, _sigma_of(s)), **extra_args)
        sd, su = get_ancestral_step(_sigma_of(t), _sigma_of(t_next), eta)
        action = to(action, (1 - fac) * denoised + fac * denoised_2, t, _neg_log(sd))
        action = action + noise_sampler(_sigma_of(t), _sigma_of(t_next)) * (s_noise * float(su))
        if scaler is not None:
            action = scaler.clip_output(action)
    return action


def _out_of_scope(name, why):
    def fn(*a, **k):
        raise NotImplementedError(f"{name} is outside the MI355X hot-path scope ({why}); see DESIGN.md")
    fn.__name__ = name
    return fn


# the reference's sample_dpmpp_2m_sde cannot run (it reads undefined names, gc_sampling.py:817-820) and nothing
# dispatches to it (beso_agent.py:452 maps 'dpmpp_2m_sde' to sample_dpmpp_sde): there is no behaviour to reproduce
sample_dpmpp_2m_sde = _out_of_scope('sample_dpmpp_2m_sde', 'the reference function raises NameError; use sample_dpmpp_sde')
log_likelihood = _out_of_scope('log_likelihood', 'needs the torchdiffeq ODE integrator')
