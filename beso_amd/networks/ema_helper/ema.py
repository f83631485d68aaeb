"""Exponential moving average of the score-network parameters (reference:
beso/networks/ema_helper/ema.py:10-104): same warm-up rule ``min(decay, (1+n)/(10+n))`` and the same
store / copy_to / restore surface.  The shadow lives in ONE flat fp32 buffer so that an update is a
single fused ``lerp`` over all parameters instead of one launch per tensor, and a ``version`` counter
lets the agent keep a packed kernel image of the shadow and refresh it only after an update."""
import torch


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, device: str = 'cuda', use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError('Decay must be between 0 and 1')
        self.decay = decay
        self._device = device
        self.num_updates = 0 if use_num_updates else None
        params = [p for p in parameters if p.requires_grad]
        self._numels = [p.numel() for p in params]
        self._flat = (torch.cat([p.detach().reshape(-1) for p in params]).clone() if params
                      else torch.zeros(0))
        self.shadow_params = self._views(self._flat, params)
        self.collected_params = []
        self.steps = 0
        self.version = 0

    def _views(self, flat, like):
        out, off = [], 0
        for p in like:
            n = p.numel()
            out.append(flat[off:off + n].view(p.shape))
            off += n
        return out

    def next_decay(self) -> float:
        """Counts one update and returns the decay it uses: min(decay, (1+n)/(10+n))   (ema.py:45-48)."""
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        return decay

    def update(self, parameters):
        """shadow -= (1 - decay) * (shadow - param)   (ema.py:36-53)."""
        decay = self.next_decay()
        with torch.no_grad():
            params = [p for p in parameters if p.requires_grad]
            if params:
                flat = torch.cat([p.detach().reshape(-1) for p in params])
                self._flat.sub_((1.0 - decay) * (self._flat - flat))
        self.version += 1

    def copy_to(self, parameters):
        for s, p in zip(self.shadow_params, [p for p in parameters if p.requires_grad]):
            p.data.copy_(s.data)

    def store(self, parameters):
        self.collected_params = [p.clone() for p in parameters]

    def restore(self, parameters):
        for c, p in zip(self.collected_params, parameters):
            p.data.copy_(c.data)

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_shadow_params(self, parameters):
        for s, p in zip(self.shadow_params, [p for p in parameters if p.requires_grad]):
            s.data.copy_(p.data)
        self.version += 1

    def load_state_dict(self, state_dict):
        self.decay = state_dict['decay']
        self.num_updates = state_dict['num_updates']
        for s, new in zip(self.shadow_params, state_dict['shadow_params']):
            s.data.copy_(new.data)
        self.version += 1
