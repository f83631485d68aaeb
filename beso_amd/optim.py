"""Fused optimizer step for the score-matching training step (kernels K11/K12).

``FusedAdam`` is a ``torch.optim.Optimizer`` with the constructor and ``param_groups`` of
``torch.optim.Adam`` / ``torch.optim.AdamW`` (the two optimizers the reference configures:
configs/agents/beso_kitchen.yaml:9-12, beso_block_push.yaml:9-11), so LR schedulers attach to it
unchanged.  ``step()`` updates ALL parameters -- and, when an EMA helper is handed over, its shadow copy
(ema.py:45-53) -- in one HIP launch (``beso_adam_step``) instead of several hundred eager launches.
There is no CPU implementation: ``maybe_fuse`` leaves a CPU optimizer untouched.  The moments live in flat
buffers owned by the optimizer object (not in ``Optimizer.state``): like the reference's training loop
(``beso_agent.py:466-476`` stores model weights only) optimizer state is not checkpointed."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

CHUNK = 4096


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled_weight_decay=False):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("invalid Adam hyper-parameter")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                        decoupled_weight_decay=decoupled_weight_decay)
        super().__init__(params, defaults)
        self._groups = [None] * len(self.param_groups)      # per group: flat state + chunk table

    @classmethod
    def from_torch(cls, opt: torch.optim.Optimizer) -> "FusedAdam":
        """Same parameters and hyper-parameters as a freshly constructed torch Adam / AdamW."""
        decoupled = isinstance(opt, torch.optim.AdamW)
        groups = [dict(params=g["params"], lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"],
                       decoupled_weight_decay=decoupled) for g in opt.param_groups]
        return cls(groups)

    def _prepare(self, gi: int, group: dict, shard=None):
        params = [p for p in group["params"] if p.grad is not None]
        sig = (shard,) + tuple((p.data_ptr(), p.grad.data_ptr(), p.numel()) for p in params)
        st = self._groups[gi]
        if st is not None and st["sig"] == sig:
            return st
        all_params = list(group["params"])
        dev = all_params[0].device
        for p in all_params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev or not p.is_cuda:
                raise ValueError("FusedAdam needs contiguous fp32 parameters on one HIP device")
        offs, total = {}, 0
        for p in all_params:                                  # state offsets are fixed by the parameter order
            offs[id(p)] = total
            total += p.numel()
        if st is None:
            st = dict(m=torch.zeros(total, device=dev), v=torch.zeros(total, device=dev), step=0, offs=offs, total=total)
        rows = []
        for p in params:
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise ValueError("FusedAdam needs contiguous fp32 gradients")
            n, off = p.numel(), offs[id(p)]
            # shard = [lo, hi) of the flat parameter order (ZeRO-1 style data parallelism: this rank owns and updates
            # only that range; the moments of the rest are never touched here)
            s0, s1 = (0, n) if shard is None else (max(0, shard[0] - off), min(n, shard[1] - off))
            for s in range(s0, s1, CHUNK):
                c = min(CHUNK, s1 - s)
                rows.append((p.data_ptr() + 4 * s, g.data_ptr() + 4 * s, off + s, c))
        st["table"] = torch.tensor(rows, dtype=torch.int64).to(dev) if rows else None      # 32-byte beso_optim_chunk rows
        st["n_chunks"] = len(rows)
        st["sig"] = sig
        self._groups[gi] = st
        return st

    @torch.no_grad()
    def step(self, closure=None, ema=None, shard=None):
        """One step.  ``ema``: an ``ExponentialMovingAverage`` over exactly this optimizer's parameters (in
        order) whose shadow is updated in the same launch, with its own warm-up rule.  ``shard = (lo, hi)``: update
        only the elements [lo, hi) of the flat parameter order (one parameter group) -- the sharded data-parallel
        step, where this rank holds the reduced gradients of that range only."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if shard is not None and len(self.param_groups) != 1:
            raise ValueError("a sharded step needs one parameter group")
        shard = None if shard is None else (int(shard[0]), int(shard[1]))
        prepared = [self._prepare(gi, group, shard) for gi, group in enumerate(self.param_groups)]   # validates devices/dtypes
        lib = _lib.load()
        ema_decay, ema_ptr = 0.0, None
        if ema is not None:
            if len(self.param_groups) != 1 or ema._flat.numel() != sum(p.numel() for p in self.param_groups[0]["params"]):
                raise ValueError("fused EMA needs one parameter group that matches the EMA helper")
            ema_decay = ema.next_decay()
            ema_ptr = ema._flat.data_ptr()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for st, group in zip(prepared, self.param_groups):
            st["step"] += 1
            if st["n_chunks"] == 0:
                continue
            b1, b2 = group["betas"]
            _lib.check(lib.beso_adam_step(st["table"].data_ptr(), st["n_chunks"], st["m"].data_ptr(), st["v"].data_ptr(),
                                          ema_ptr, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                          float(group["weight_decay"]), 1 if group["decoupled_weight_decay"] else 0,
                                          st["step"], float(ema_decay), stream), "beso_adam_step")
        # the kernel writes the parameters through raw pointers: bump their version counters so that everything keyed on
        # them (the packed-weight cache of DiffusionGPT, autograd's saved-tensor checks) sees the update
        torch.autograd.graph.increment_version([p for g in self.param_groups for p in g["params"] if p.grad is not None])
        if ema is not None:
            ema.version += 1
        return loss


def maybe_fuse(opt: torch.optim.Optimizer) -> torch.optim.Optimizer:
    """torch Adam / AdamW over HIP fp32 parameters with default flags -> FusedAdam; anything else unchanged."""
    if type(opt) not in (torch.optim.Adam, torch.optim.AdamW):
        return opt
    for g in opt.param_groups:
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            return opt
        if type(opt) is torch.optim.Adam and g.get("decoupled_weight_decay"):
            return opt
        for p in g["params"]:
            if not p.is_cuda or p.dtype != torch.float32:
                return opt
    if any(len(s) for s in opt.state.values()):
        return opt                                             # already stepped: keep its state
    return FusedAdam.from_torch(opt)
