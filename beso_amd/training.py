"""HIP training step of the score network: ``GCDenoiser.loss`` forward and the
gradient of the loss with respect to every parameter as ONE enqueue of ``beso_loss_grad``
(beso_amd/csrc/train.hip), replacing ``loss = model.loss(...); loss.backward()`` of the reference's
``train_step`` (beso_agent.py:226-235).

Two ways in:

* ``GCDenoiser.loss`` (training mode, HIP parameters) returns ``ScoreMatchingLoss.apply(...)``, an autograd
  node whose backward hands the precomputed gradients to autograd -- user code that calls ``loss.backward()``
  keeps working unchanged;
* ``HipTrainStep.loss_backward`` is what ``BesoAgent.train_step`` uses: it writes ``p.grad`` as views of one
  persistent flat buffer (stable addresses for the fused optimizer's chunk table, one flat tensor for the
  data-parallel all-reduce) and returns the loss.

There is no CPU implementation and no second evaluation of the network behind it: what the kernels do not cover
(``embed_dim`` not a multiple of 8, attention windows beyond the LDS budget, unknown ``loss`` kwargs) raises
(``GCDenoiser.loss``); the torch-autograd evaluation of the same function lives in ``tests/autograd_reference.py`` as
the comparator of this step.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib
from .runtime import train_hints


class HipTrainStep:
    """Binds one ``DiffusionGPT`` to ``beso_loss_grad``: flat gradient buffer, workspace, dropout seed."""

    def __init__(self, inner, sigma_data: float):
        self.inner = inner
        self.sigma_data = float(sigma_data)
        self.lib = _lib.load()
        self.cfg = inner.shape(self.sigma_data).c_struct()
        self.n_params = self.lib.beso_num_params(C.byref(self.cfg))
        self.n_grad = int(self.lib.beso_grad_floats(C.byref(self.cfg)))
        self._flat: Optional[torch.Tensor] = None
        self._flat_full: Optional[torch.Tensor] = None      # _flat plus the zero tail of the sharded exchange
        self.pad_to = 0
        self._views: Optional[List[torch.Tensor]] = None
        self._ws: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ eligibility
    @staticmethod
    def supported(inner) -> bool:
        return inner.embed_dim % 8 == 0

    def _params(self):
        """(Parameter list, their data pointers as a tuple).  Walking the module tree costs ~50 us per traversal and a step
        used to do four; the list is kept (a module's Parameter objects do not change identity under .to(), load_state_dict
        or the EMA swap -- their storage may, which the pointer tuple shows) and rebuilt when its length stops matching."""
        pl = self.__dict__.get("_plist")
        if pl is None or len(pl) != self.n_params or self.__dict__.get("_plist_epoch") != getattr(self.inner, "_dirty_epoch", 0):
            pl = list(self.inner.parameters())
            self._plist, self._plist_epoch = pl, getattr(self.inner, "_dirty_epoch", 0)
            self._ptrs_ok = None
        return pl, tuple(p.data_ptr() for p in pl)

    def eligible(self, state, action, goal, noise, sigma) -> bool:
        params, ptrs = self._params()
        if self.__dict__.get("_ptrs_ok") != ptrs:             # (checked once per set of storages)
            if not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.requires_grad for p in params):
                return False
            if sum(p.numel() for p in params) != self.n_grad or len(params) != self.n_params:
                return False
            self._ptrs_ok = ptrs
        elif not all(p.requires_grad for p in params):
            return False
        for x in (state, action, noise, sigma) + ((goal,) if goal is not None else ()):
            if not torch.is_tensor(x) or not x.is_cuda or x.requires_grad:
                return False
        if self.inner.goal_seq_len > 0 and goal is None:
            return False
        return self.supported(self.inner)

    # ------------------------------------------------------------------ buffers
    def _grad_buffer(self, dev, fresh: bool):
        if fresh:
            flat = torch.empty(self.n_grad, dtype=torch.float32, device=dev)
        else:
            if (self._flat is None or self._flat_full is None or self._flat.device != dev
                    or self._flat_full.numel() != max(self.n_grad, self.pad_to)):
                # (pad_to: the sharded data-parallel exchange reduce-scatters a buffer of world * ceil(n / world) floats;
                # the kernels write the first n_grad, the tail stays zero)
                self._flat_full = torch.zeros(max(self.n_grad, self.pad_to), dtype=torch.float32, device=dev)
                self._flat = self._flat_full[: self.n_grad]
                self._views = None
            flat = self._flat
        views, off = [], 0
        if fresh or self._views is None:
            for p in self._params()[0]:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            if not fresh:
                self._views = views
        else:
            views = self._views
        return flat, views

    def _workspace(self, batch: int, t: int, precision: int, dev) -> torch.Tensor:
        need = int(self.lib.beso_train_workspace_bytes(C.byref(self.cfg), batch, t, precision))
        if need == 0:
            raise ValueError(f"beso_hip: training step unsupported for batch={batch} t={t} with this model shape")
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws

    # ------------------------------------------------------------------ the call
    def early_range(self):
        """[begin, end) floats of the flat gradient buffer that are complete first (the upper layers + ln_f):
        what ``run(..., early_stream=...)`` releases to that stream while the rest of the backward still runs."""
        b, e = C.c_size_t(0), C.c_size_t(0)
        _lib.check(self.lib.beso_grad_early_range(C.byref(self.cfg), C.byref(b), C.byref(e)), "grad_early_range")
        return int(b.value), int(e.value)

    def run(self, state, action, goal, noise, sigma, grad_scale: float = 1.0, seed: Optional[int] = None,
            fresh_grads: bool = False, last_action_only: bool = False, early_stream=None, goal_drop: Optional[float] = None,
            loss_stream=None):
        """-> (loss 0-d tensor, flat gradient tensor, list of per-parameter views).  Inputs are NOT modified.
        ``early_stream`` (a torch.cuda.Stream): ordered behind the completion of ``early_range()`` by the call.
        ``loss_stream`` (a torch.cuda.Stream): ordered behind the point where the loss value is final (the end of the forward
        half) -- a host read of the loss on that stream does not wait for the backward pass.
        ``goal_drop``: None = the module's ``cond_mask_prob`` (training mode); 0 = the goals are taken as they are."""
        inner = self.inner
        dev = action.device
        f32 = lambda x: x.detach().to(device=dev, dtype=torch.float32).contiguous()
        state, action, noise, sigma = f32(state), f32(action), f32(noise), f32(sigma).reshape(-1)
        if state.dim() != 3 or action.dim() != 3:
            raise ValueError("state must be [B,t,obs] and action [B,t,act]")
        B, t, _ = state.shape
        if action.shape[:2] != (B, t) or noise.shape != action.shape or sigma.numel() != B:
            raise ValueError("action / noise must be [B,t,act] and sigma [B]")
        G = inner.goal_seq_len
        gptr = None
        if G > 0:
            goal = f32(goal)
            if goal.dim() == 2:
                goal = goal.unsqueeze(0)
            goal = goal.expand(B, G, inner.obs_dim).contiguous()
            gptr = goal.data_ptr()
        embed_p, attn_p, resid_p = inner._pdrops
        # training-mode goal masking (DiffusionGPT.mask_cond, score_gpts.py:298-299) is part of the kernel: the goals
        # go in unmasked together with the drop probability
        goal_p = float(getattr(inner, "cond_mask_prob", 0.0) or 0.0) if (G > 0 and goal_drop is None) else float(goal_drop or 0.0)
        if not inner.training:
            embed_p = attn_p = resid_p = goal_p = 0.0
        if seed is None:
            if (embed_p > 0.0 or attn_p > 0.0 or resid_p > 0.0 or goal_p > 0.0) and torch.cuda.is_current_stream_capturing():
                # the seed is a host scalar: a captured graph would replay one mask forever
                raise RuntimeError("beso_amd: the HIP training step with dropout cannot be captured into a graph")
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,), device="cpu").item())
        # the split-bf16 and fp16 modes are inference instances of the fused kernel; their training steps are the fp32 / bf16 ones
        precision = _lib.PRECISIONS[{"bf16x3": "fp32", "fp16": "bf16"}.get(inner.precision, inner.precision)]
        params, ptrs = self._params()
        if self.__dict__.get("_arr_ptrs") != ptrs:
            self._arr, self._arr_ptrs = (C.c_void_p * len(ptrs))(*ptrs), ptrs
        arr = self._arr
        flat, views = self._grad_buffer(dev, fresh_grads)
        ws = self._workspace(B, t, precision, dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        if loss_stream is not None:
            loss.record_stream(loss_stream)          # (read there; the caching allocator must not hand it out before that)
        common = (C.byref(self.cfg), arr, len(params), flat.data_ptr(), precision,
                  state.data_ptr(), action.data_ptr(), gptr, noise.data_ptr(), sigma.data_ptr(),
                  loss.data_ptr(), B, t, (_lib.TRAIN_LAST_ACTION_ONLY if last_action_only else 0) | train_hints(), float(embed_p), float(attn_p), float(resid_p),
                  float(goal_p), C.c_uint(seed & 0xFFFFFFFF), float(grad_scale), ws.data_ptr(), ws.numel(),
                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream),
                  C.c_void_p(early_stream.cuda_stream) if early_stream is not None else None)
        with torch.cuda.device(dev):
            if hasattr(self.lib, "beso_loss_grad_streams"):
                st = self.lib.beso_loss_grad_streams(*common, C.c_void_p(loss_stream.cuda_stream) if loss_stream is not None else None)
            else:
                # (a library loaded through BESO_HIP_LIB that predates the loss stream -- tools/variants.py's '@<revision>' A/B
                # builds: the loss is then final at the end of the call's work on the compute stream; order the caller's stream
                # behind that instead of failing)
                st = self.lib.beso_loss_grad_overlap(*common)
                if loss_stream is not None:
                    loss_stream.wait_stream(torch.cuda.current_stream(dev))
        _lib.check(st, "loss_grad")
        return loss, flat, views

    def goal_mask(self, batch: int, seed: int, goal_drop: Optional[float] = None, device=None) -> torch.Tensor:
        """The keep-mask [batch, G, obs] the kernel applies to the goals for (goal_drop, seed) (``beso_goal_mask``)."""
        inner = self.inner
        p = float(inner.cond_mask_prob if goal_drop is None else goal_drop)
        dev = torch.device(device) if device is not None else next(inner.parameters()).device
        mask = torch.empty(batch, inner.goal_seq_len, inner.obs_dim, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(self.lib.beso_goal_mask(mask.data_ptr(), batch, inner.goal_seq_len, inner.obs_dim, p,
                                               C.c_uint(seed & 0xFFFFFFFF),
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "goal_mask")
        return mask

    def loss_backward(self, state, action, goal, noise, sigma, grad_scale: float = 1.0, seed: Optional[int] = None,
                      early_stream=None, loss_stream=None):
        """The training step's ``loss = model.loss(...); loss.backward()``: returns the loss and leaves the
        gradients in ``p.grad`` (views of the persistent flat buffer; accumulated into an existing ``.grad``)."""
        loss, flat, views = self.run(state, action, goal, noise, sigma, grad_scale, seed, fresh_grads=False,
                                     early_stream=early_stream, loss_stream=loss_stream)
        for p, v in zip(self._params()[0], views):
            if p.grad is None or p.grad.data_ptr() == v.data_ptr():
                p.grad = v
            else:
                p.grad.add_(v)
        return loss

    def flat_grads_padded(self) -> Optional[torch.Tensor]:
        """flat_grads() with its zero tail (``pad_to`` floats in all), for the reduce-scatter of the sharded exchange."""
        return None if self.flat_grads() is None else self._flat_full

    def flat_grads(self) -> Optional[torch.Tensor]:
        """The flat buffer if every ``p.grad`` currently is its view of it (then one all-reduce covers them)."""
        if self._flat is None or self._views is None:
            return None
        for p, v in zip(self._params()[0], self._views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                return None
        return self._flat


class ScoreMatchingLoss(torch.autograd.Function):
    """loss = GCDenoiser.loss(...) with the parameter gradients computed alongside (HIP); backward hands them over."""

    @staticmethod
    def forward(ctx, step: HipTrainStep, last_action_only: bool, state, action, goal, noise, sigma, *params):
        loss, flat, views = step.run(state, action, goal, noise, sigma, fresh_grads=True, last_action_only=last_action_only)
        ctx.flat, ctx.views = flat, views
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        scaled = ctx.flat * grad_out               # out of place: a second backward through a retained graph stays right
        grads, off = [], 0
        for v in ctx.views:
            grads.append(scaled[off:off + v.numel()].view_as(v))
            off += v.numel()
        return (None,) * 7 + tuple(grads)
