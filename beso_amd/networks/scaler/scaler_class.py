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
