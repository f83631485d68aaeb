"""Boundary plumbing: z-score scaler of observations / actions with the action bounds the agent
clips to (reference: beso/networks/scaler/scaler_class.py:11-166).  Elementwise host-side torch;
the workspace normally builds the reference's own Scaler -- any object with this surface works."""
import numpy as np
import torch


class Scaler:
    def __init__(self, x_data, y_data, scale_data: bool, device: str):
        self.scale_data = scale_data
        self.device = device
        if isinstance(x_data, torch.Tensor):
            x_data, y_data = x_data.detach().cpu().numpy(), y_data.detach().cpu().numpy()
        if x_data.ndim == 3:
            x_data = x_data.reshape(-1, x_data.shape[-1])
            y_data = y_data.reshape(-1, y_data.shape[-1])
        elif x_data.ndim not in (2, 4):
            raise ValueError('not implemented yet!')
        dev = lambda a: torch.from_numpy(a).to(device)       # noqa: E731
        self.x_mean, self.x_std = dev(x_data.mean(0)), dev(x_data.std(0))
        self.y_mean, self.y_std = dev(y_data.mean(0)), dev(y_data.std(0))
        self.x_max, self.x_min = dev(x_data.max(0)), dev(x_data.min(0))
        self.y_max, self.y_min = dev(y_data.max(0)), dev(y_data.min(0))
        self.y_bounds = np.zeros((2, y_data.shape[-1]))
        self.x_bounds = np.zeros((2, x_data.shape[-1]))
        if scale_data:
            self.y_bounds[0] = (y_data.min(0) - y_data.mean(0)) / (y_data.std(0) + 1e-12)
            self.y_bounds[1] = (y_data.max(0) - y_data.mean(0)) / (y_data.std(0) + 1e-12)
            self.x_bounds[0] = (x_data.min(0) - x_data.mean(0)) / (x_data.std(0) + 1e-12)
            self.x_bounds[1] = (x_data.max(0) - x_data.mean(0)) / (x_data.std(0) + 1e-12)
        else:
            self.y_bounds[0], self.y_bounds[1] = y_data.min(0), y_data.max(0)
            self.x_bounds[0], self.x_bounds[1] = x_data.min(0), x_data.max(0)
        self.y_bounds_tensor = torch.from_numpy(self.y_bounds).to(device)
        self.x_bounds_tensor = torch.from_numpy(self.x_bounds).to(device)
        self.tensor_y_bounds = self.y_bounds_tensor

    @torch.no_grad()
    def scale_input(self, x):
        x = x.to(self.device)
        if x.shape[-1] == 7 and len(self.x_mean) == 30:      # one-hot kitchen goals pass through
            return x
        if self.scale_data:
            return ((x - self.x_mean) / self._den("x")).to(torch.float32)
        return x

    @torch.no_grad()
    def scale_output(self, y):
        y = y.to(self.device)
        if self.scale_data:
            return ((y - self.y_mean) / self._den("y")).to(torch.float32)
        return y

    @torch.no_grad()
    def scale_many(self, items):
        """[(tensor, 'x' | 'y'), ...] -> the scaled tensors: scale_input / scale_output of every item (scaler_class.py:95-117) as
        ONE HIP launch (beso_scale_rows: the same subtraction and correctly rounded division per element) where every tensor
        is a contiguous fp32 CUDA tensor with fp32 statistics -- a training batch is three tensors, i.e. six elementwise launches
        of ~6 us each in front of the forward.  Anything else (CPU tensors, other dtypes, one-hot goals, scale_data off, a
        library without the entry point): the per-tensor methods."""
        plan, ok = [], self.scale_data and 0 < len(items) <= 4
        for x, which in items:
            mean = self.x_mean if which == "x" else self.y_mean
            den = self._den(which) if self.scale_data else None
            good = (ok and isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.numel() > 0
                    and isinstance(mean, torch.Tensor) and mean.is_cuda and mean.device == x.device and mean.dtype == torch.float32
                    and den.dtype == torch.float32 and mean.dim() == 1 and mean.numel() == x.shape[-1] and mean.is_contiguous()
                    and den.is_contiguous() and not (which == "x" and x.shape[-1] == 7 and len(self.x_mean) == 30))
            ok = ok and good
            plan.append((x, mean, den))
        lib = None
        if ok:
            from ... import _lib
            lib = _lib.load()
            ok = hasattr(lib, "beso_scale_rows")
        if not ok:
            return [self.scale_input(x) if which == "x" else self.scale_output(x) for x, which in items]
        import ctypes as C
        from ... import _lib
        n = len(plan)
        outs = [torch.empty_like(x) for x, _, _ in plan]
        vp = C.c_void_p * n
        src = vp(*[x.data_ptr() for x, _, _ in plan]); dst = vp(*[o.data_ptr() for o in outs])
        mean = vp(*[m.data_ptr() for _, m, _ in plan]); den = vp(*[d.data_ptr() for _, _, d in plan])
        rows = (C.c_longlong * n)(*[x.numel() // x.shape[-1] for x, _, _ in plan])
        cols = (C.c_int * n)(*[x.shape[-1] for x, _, _ in plan])
        dev = plan[0][0].device
        with torch.cuda.device(dev):
            _lib.check(lib.beso_scale_rows(src, dst, mean, den, rows, cols, n, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                       "scale_rows")
        return outs

    def _den(self, which: str):
        """std + 1e-12 (scaler_class.py:129,139): the same tensor every call -- computed once per std tensor instead of one
        more elementwise launch per scaled batch (three per training step)."""
        std = self.x_std if which == "x" else self.y_std
        cache = self.__dict__.setdefault("_den_cache", {})
        hit = cache.get(which)
        if hit is None or hit[0] is not std or hit[1] != std._version:
            hit = (std, std._version, std + 1e-12)
            cache[which] = hit
        return hit[2]

    @torch.no_grad()
    def inverse_scale_input(self, x):
        return x * (self.x_std + 1e-12) + self.x_mean if self.scale_data else x

    @torch.no_grad()
    def inverse_scale_output(self, y):
        y = y.to(self.device)
        return y * self._den("y") + self.y_mean if self.scale_data else y

    @torch.no_grad()
    def clip_action(self, y):
        """Clamp to 1.1 x the data bounds (scaler_class.py:162-166).  The two scaled bound rows are the same tensors every
        call: computed once per bounds tensor (two elementwise launches less per environment step)."""
        b = self.y_bounds_tensor
        hit = self.__dict__.get("_clip_cache")
        if hit is None or hit[0] is not b or hit[1] != b._version:
            hit = (b, b._version, b[0] * 1.1, b[1] * 1.1)
            self._clip_cache = hit
        return torch.clamp(y, hit[2], hit[3]).to(self.device).to(torch.float32)
