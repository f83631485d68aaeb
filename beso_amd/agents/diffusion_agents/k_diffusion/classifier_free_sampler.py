"""Drop-in for ``beso...k_diffusion.classifier_free_sampler.ClassifierFreeSampleModel``
(reference: classifier_free_sampler.py:12-52): classifier-free guidance at SAMPLING time,
``out_uncond + lambda * (out_cond - out_uncond)``.

Around a ``beso_amd`` GCDenoiser the conditional and unconditional passes run as one 2B-sample
launch sequence inside the HIP library (the second half of the batch embeds zero goals)."""
from copy import deepcopy

import torch.nn as nn

from .score_wrappers import GCDenoiser


class ClassifierFreeSampleModel(nn.Module):
    def __init__(self, model, cond_lambda: float = 2):
        super().__init__()
        self.model = model
        self.cond_lambda = cond_lambda
        self.cond = bool(cond_lambda == 1)       # lambda == 1: purely conditional (:30-33)

    def forward(self, state, action, goal, sigma, **extra_args):
        if self.cond:
            return self.model(state, action, goal, sigma)
        if self.cond_lambda == 0:
            return self.model(state, action, goal, sigma, uncond=True)
        m = self.model
        if isinstance(m, GCDenoiser) and not extra_args and m._fused(m.inner_model, {}, state, action, goal, sigma):
            inner = m.inner_model
            return inner.runtime(m.sigma_data).denoise(inner.packed_weights(), state, action, goal, sigma,
                                                       cond_lambda=float(self.cond_lambda), precondition=True)
        action = deepcopy(action)
        out = m(state, action, goal, sigma, **extra_args)        # extra args reach the conditional call only (:45-47)
        out_uncond = m(state, action, goal, sigma, uncond=True)
        return out_uncond + self.cond_lambda * (out - out_uncond)

    def get_params(self):
        return self.model.get_params()
