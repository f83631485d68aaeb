"""Helpers of the k-diffusion layer that sit on the hot path (reference: k_diffusion/utils.py:165-220):
``append_dims`` and the sigma training densities.  The five alternative time-embedding classes of
the reference are out of scope -- DiffusionGPT ignores ``time_embedding_fn`` (score_gpts.py:136,178)."""
import math

import numpy as np
import torch


def append_dims(x, target_dims):
    """Trailing singleton dims until ``x.ndim == target_dims`` (utils.py:165-170)."""
    missing = target_dims - x.ndim
    if missing < 0:
        raise ValueError(f'input has {x.ndim} dims but target_dims is {target_dims}, which is less')
    return x[(...,) + (None,) * missing]


def rand_log_normal(shape, loc=0., scale=1., device='cpu', dtype=torch.float32):
    """exp(N(loc, scale))  (utils.py:173-175)."""
    return torch.randn(shape, device=device, dtype=dtype).mul_(scale).add_(loc).exp_()


def _hip_lib():
    from .... import _lib
    return _lib.load()


def rand_log_logistic(shape, loc=0., scale=1., min_value=0., max_value=float('inf'), device='cpu',
                      dtype=torch.float32):
    """Truncated log-logistic, drawn in float64 like the reference (utils.py:178-185).  The two CDF bounds
    are host doubles (the reference builds them as 0-d device tensors from the same Python floats): no
    host-to-device copy, so the draw can be captured into a HIP graph."""
    def cdf(v):
        if v <= 0.0:
            return 0.0
        z = (math.log(v) - loc) / scale if v != float('inf') else float('inf')
        return 1.0 / (1.0 + math.exp(-z)) if z > -700.0 else 0.0
    lo, hi = cdf(float(min_value)), cdf(float(max_value))
    u = torch.rand(shape, device=device, dtype=torch.float64)
    if u.is_cuda and dtype == torch.float32 and hasattr(_hip_lib(), "beso_log_logistic"):      # (older A/B libraries: the chain below)
        # the transform behind the draw as ONE HIP launch (beso_log_logistic: the same float64 operations in the same order)
        # instead of seven elementwise ones -- the training step draws its sigmas here every step (beso_agent.py:227)
        import ctypes as C
        from .... import _lib
        out = torch.empty(u.shape, device=u.device, dtype=torch.float32)
        with torch.cuda.device(u.device):
            _lib.check(_lib.load().beso_log_logistic(u.data_ptr(), out.data_ptr(), u.numel(), float(loc), float(scale), lo, hi,
                                                     C.c_void_p(torch.cuda.current_stream(u.device).cuda_stream)), "log_logistic")
        return out
    return (u * (hi - lo) + lo).logit().mul(scale).add(loc).exp().to(dtype)


def rand_log_uniform(shape, min_value, max_value, device='cpu', dtype=torch.float32):
    lo, hi = math.log(min_value), math.log(max_value)
    return (torch.rand(shape, device=device, dtype=dtype) * (hi - lo) + lo).exp()


def rand_uniform(shape, min_value, max_value, device='cpu', dtype=torch.float32):
    return torch.rand(shape, device=device, dtype=dtype) * (max_value - min_value) + min_value


def rand_discrete(shape, values, device='cpu', dtype=torch.float32):
    values = values.detach().cpu().numpy() if torch.is_tensor(values) else np.asarray(values)
    return torch.tensor(np.random.choice(values, size=shape), device=device, dtype=dtype)


def rand_v_diffusion(shape, sigma_data=1., min_value=0., max_value=float('inf'), device='cpu', dtype=torch.float32):
    lo = math.atan(min_value / sigma_data) * 2 / math.pi
    hi = math.atan(max_value / sigma_data) * 2 / math.pi
    u = torch.rand(shape, device=device, dtype=dtype) * (hi - lo) + lo
    return torch.tan(u * math.pi / 2) * sigma_data


def rand_split_log_normal(shape, loc, scale_1, scale_2, device='cpu', dtype=torch.float32):
    n = torch.randn(shape, device=device, dtype=dtype).abs()
    u = torch.rand(shape, device=device, dtype=dtype)
    left, right = n * -scale_1 + loc, n * scale_2 + loc
    return torch.where(u < scale_1 / (scale_1 + scale_2), left, right).exp()
