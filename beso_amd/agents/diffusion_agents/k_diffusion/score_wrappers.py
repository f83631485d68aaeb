"""Drop-in for ``beso...k_diffusion.score_wrappers.GCDenoiser``: the Karras et al. (2022)
preconditioner around the score transformer (reference: score_wrappers.py:18-99).

With a ``beso_amd`` DiffusionGPT inside, ``forward`` is ONE call into the HIP library
(``beso_denoise_fwd``): c_in is folded into the token-embedding kernel, c_out / c_skip into the
action-head kernel.  Any other inner model is evaluated by the textbook formula on top of it.
"""
import torch
from torch import nn

from ...._instantiate import instantiate
from .score_gpts import DiffusionGPT
from .utils import append_dims
from ....training import HipTrainStep, ScoreMatchingLoss


class GCDenoiser(nn.Module):
    """D_theta(a; s, g, sigma) = c_skip * a + c_out * F(s, c_in * a, g, sigma)."""

    def __init__(self, inner_model, sigma_data=1.):
        super().__init__()
        self.inner_model = inner_model if isinstance(inner_model, nn.Module) else instantiate(inner_model)
        self.sigma_data = sigma_data
        self._train_steps = {}

    # -- reference surface ---------------------------------------------------------------------
    def get_scalings(self, sigma):
        """(c_skip, c_out, c_in) of score_wrappers.py:40-42."""
        sd2 = self.sigma_data ** 2
        total = sigma ** 2 + sd2
        return sd2 / total, sigma * self.sigma_data / total ** 0.5, 1 / total ** 0.5

    def get_params(self):
        return self.inner_model.parameters()

    def forward(self, state, action, goal, sigma, **kwargs):
        inner = self.inner_model
        if self._fused(inner, kwargs, state, action, goal, sigma):
            out = inner.runtime(self.sigma_data).denoise(
                inner.packed_weights(), state, action, goal, sigma,
                uncond=bool(kwargs.get("uncond", False)), precondition=True)
            return out
        c_skip, c_out, c_in = (append_dims(c, action.ndim) for c in self.get_scalings(sigma))
        return inner(state, action * c_in, goal, sigma, **kwargs) * c_out + action * c_skip

    def loss(self, state, action, goal, noise, sigma, **kwargs):
        """Score-matching objective (score_wrappers.py:45-79).  Mutates ``noise`` in place when
        ``pred_last_action_only`` is set, like the reference (:63)."""
        last_only = bool(kwargs.pop("pred_last_action_only", False))
        if last_only:
            noise[:, :-1, :] = 0                                   # in place, like the reference (:63)
        inner = self.inner_model
        if isinstance(inner, DiffusionGPT):
            # forward + every parameter gradient in one enqueue of beso_loss_grad (beso_amd/training.py); there is no
            # torch-op evaluation behind it: what the kernels cannot serve raises
            if kwargs:
                raise ValueError(f"GCDenoiser.loss: unsupported keyword arguments {sorted(kwargs)} for the HIP training step")
            step = self.hip_train_step(state, action, goal, noise, sigma)
            if step is None:
                raise ValueError("GCDenoiser.loss: " + self._why_no_hip_step(state, action, goal, noise, sigma))
            # (training-mode goal masking -- DiffusionGPT.mask_cond, score_gpts.py:298-299 -- happens inside the kernel)
            return ScoreMatchingLoss.apply(step, last_only, state, action, goal, noise, sigma, *inner.parameters())
        # a foreign inner model (any callable score network): the textbook objective on top of it
        noised = action + noise * append_dims(sigma, action.ndim)
        c_skip, c_out, c_in = (append_dims(c, action.ndim) for c in self.get_scalings(sigma))
        out = inner(state, noised * c_in, goal, sigma, **kwargs)
        target = (action - c_skip * noised) / c_out
        if last_only:
            return (out[:, -1, :] - target[:, -1, :]).pow(2).mean()
        return (out - target).pow(2).flatten(1).mean()

    # -- HIP training step ---------------------------------------------------------------------
    def hip_train_step(self, state, action, goal, noise, sigma):
        """The ``HipTrainStep`` bound to the inner model if this call can run on it, else None."""
        inner = self.inner_model
        if not isinstance(inner, DiffusionGPT) or not torch.is_grad_enabled():
            return None
        if not HipTrainStep.supported(inner):
            return None
        if not (torch.is_tensor(action) and action.is_cuda):
            return None
        key = float(self.sigma_data)
        step = self._train_steps.get(key)
        if step is None:
            step = self._train_steps[key] = HipTrainStep(inner, key)
        return step if step.eligible(state, action, goal, noise, sigma) else None

    def _why_no_hip_step(self, state, action, goal, noise, sigma) -> str:
        inner = self.inner_model
        if not torch.is_grad_enabled():
            return "called with autograd disabled (the loss is a training-step quantity)"
        if not HipTrainStep.supported(inner):
            return f"embed_dim={inner.embed_dim} is not a multiple of 8 (the HIP training kernels need that)"
        if not (torch.is_tensor(action) and action.is_cuda):
            return "the inputs are not on the GPU (beso_amd has no CPU path)"
        return ("parameters must be contiguous fp32 HIP tensors that require grad, and state / action / goal / noise / "
                "sigma HIP tensors that do not")

    # -- fused path ----------------------------------------------------------------------------
    @staticmethod
    def _fused(inner, kwargs, *tensors) -> bool:
        if not isinstance(inner, DiffusionGPT):
            return False
        if set(kwargs) - {"uncond", "keep_last_actions"} or kwargs.get("keep_last_actions", False):
            return False
        if inner.training:
            return False            # training-mode goal masking / dropout live in DiffusionGPT.forward
        return inner._hip_eligible(*tensors)

    def fused_sampler(self, sampler: str, state, x_t, goal, sigmas, cond_lambda: float = 1.0, eta: float = 1.0, noise=None,
                      stepwise: bool = False):
        """Whole ddim / euler / heun / euler_ancestral loop as one enqueue (``beso_sample``, ``beso_sample_ancestral``); None
        if not applicable."""
        inner = self.inner_model
        if not self._fused(inner, {}, state, x_t, goal) or x_t.dim() != 3 or state.dim() != 3:
            return None
        if sampler == "euler_ancestral":
            return inner.runtime(self.sigma_data).sample_ancestral(inner.packed_weights(), state, x_t, goal, sigmas,
                                                                   cond_lambda=cond_lambda, eta=eta, noise=noise, stepwise=stepwise)
        return inner.runtime(self.sigma_data).sample(inner.packed_weights(), sampler, state, x_t, goal, sigmas,
                                                     cond_lambda=cond_lambda, stepwise=stepwise)
