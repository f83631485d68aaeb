"""Host-side runtime over the C ABI (include/beso_hip.h): packed-weight images, workspaces and the
denoise / sample calls.  PyTorch is used only for device memory and streams.

Nothing here computes the network on the CPU or with torch ops: if the HIP library is missing the
constructor raises (``beso_amd._lib.load``).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib

_hints = threading.local()


@contextlib.contextmanager
def plan(forward: int = 0, train: int = 0):
    """Execution-plan hints for the library calls the CALLING THREAD makes inside the block: ``forward`` = BESO_PLAN_* bits
    (``_lib.PLAN_PER_OP``, ``PLAN_BLOCKS``, ``PLAN_SMALL``, ``PLAN_FUSED``, ``PLAN_SPW2 / 4 / 8``) added to the flags of every forward / sampler call,
    ``train`` = ``_lib.TRAIN_PLAN_PER_OP`` / ``TRAIN_PLAN_TILES`` for ``beso_loss_grad``.  They select WHICH kernels run,
    never what is computed (parity tests: per-op kernels against the fused ones; measurements); each library call carries
    its own flags, so nothing process-wide changes."""
    old = (getattr(_hints, "forward", 0), getattr(_hints, "train", 0))
    _hints.forward, _hints.train = forward, train
    try:
        yield
    finally:
        _hints.forward, _hints.train = old


def set_plan(forward: Optional[int] = None, train: Optional[int] = None) -> None:
    """The same hints without a block: they stay with the calling thread until set again (0 = the library's own choice)."""
    if forward is not None:
        _hints.forward = forward
    if train is not None:
        _hints.train = train


def forward_hints() -> int:
    return getattr(_hints, "forward", 0)


def train_hints() -> int:
    return getattr(_hints, "train", 0)


@dataclass(frozen=True)
class ScoreNetShape:
    """The kwargs of DiffusionGPT.__init__ that shape the computation (score_gpts.py:121-139)."""
    obs_dim: int
    act_dim: int
    embed_dim: int
    n_layers: int
    n_heads: int
    goal_seq_len: int      # effective: 0 when not goal conditioned (score_gpts.py:143-144)
    obs_seq_len: int
    linear_output: bool = True
    sigma_data: float = 1.0

    def c_struct(self) -> _lib.BesoConfig:
        return _lib.BesoConfig(self.obs_dim, self.act_dim, self.embed_dim, self.n_layers, self.n_heads,
                               self.goal_seq_len, self.obs_seq_len, int(self.linear_output), float(self.sigma_data))

    def tokens(self, t: int) -> int:
        return 1 + self.goal_seq_len + 2 * t

    def flops_per_sample(self, t: Optional[int] = None) -> int:
        """Algorithmic FLOPs of one score-net forward per sample."""
        t = self.obs_seq_len if t is None else t
        D, L, G, T = self.embed_dim, self.n_layers, self.goal_seq_len, self.tokens(t)
        return (L * (24 * T * D * D + 4 * T * T * D)
                + 2 * D * (t * self.obs_dim + G * self.obs_dim + t * self.act_dim + 1) + 2 * t * D * self.act_dim)


class PackedWeights:
    """A kernel-ready image of one set of parameter values (K0 of SURVEY.md 2.1)."""

    def __init__(self, buf: torch.Tensor, precision: int, key):
        self.buf = buf
        self.precision = precision
        self.key = key


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _f32c(x: torch.Tensor, device) -> torch.Tensor:
    if x.device != device or x.dtype != torch.float32:
        x = x.to(device=device, dtype=torch.float32)
    return x.contiguous()


class ScoreNetRuntime:
    """Binds one model shape to the HIP library: pack weights, run GCDenoiser.forward /
    DiffusionGPT.forward / whole sampling loops on the current CUDA(HIP) stream."""

    def __init__(self, shape: ScoreNetShape, precision: str = "bf16"):
        self.lib = _lib.load()
        self.shape = shape
        self.cfg = shape.c_struct()
        self.set_precision(precision)
        self.n_params = self.lib.beso_num_params(C.byref(self.cfg))
        if self.n_params <= 0:
            raise ValueError(f"beso_hip: unsupported model shape {shape}")
        self._ws: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ configuration
    def set_precision(self, precision: str) -> None:
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"unknown precision {precision!r}; choose from {sorted(_lib.PRECISIONS)}")
        self.precision_name = precision
        self.precision = _lib.PRECISIONS[precision]

    # ------------------------------------------------------------------ weights
    def pack(self, params: Sequence[torch.Tensor], key=None, into: Optional[PackedWeights] = None) -> PackedWeights:
        """``params`` in the order of the reference module's ``named_parameters()``; CUDA fp32."""
        params = [p.detach() for p in params]
        if len(params) != self.n_params:
            raise ValueError(f"expected {self.n_params} parameter tensors, got {len(params)}")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("beso_amd: parameters must live on the GPU (no CPU path)")
        keep = [_f32c(p, dev) for p in params]
        nbytes = self.lib.beso_packed_bytes(C.byref(self.cfg), self.precision)
        if nbytes == 0:
            raise ValueError(f"beso_hip: precision {self.precision_name!r} unsupported for this shape")
        if into is not None and into.buf.numel() == nbytes and into.buf.device == dev and into.precision == self.precision:
            buf = into.buf
        else:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        arr = (C.c_void_p * len(keep))(*[p.data_ptr() for p in keep])
        with torch.cuda.device(dev):
            st = self.lib.beso_pack_weights(C.byref(self.cfg), arr, len(keep), buf.data_ptr(), nbytes,
                                            self.precision, _stream_ptr(dev))
        _lib.check(st, "pack_weights")
        if into is not None and buf is into.buf:
            into.key = key
            return into
        return PackedWeights(buf, self.precision, key)

    # ------------------------------------------------------------------ workspace
    def _workspace(self, batch: int, t: int, two: bool, dev) -> torch.Tensor:
        need = self.lib.beso_workspace_bytes(C.byref(self.cfg), batch, t, self.precision, int(two))
        if need == 0:
            raise ValueError(f"beso_hip: bad shape batch={batch} t={t} (t must be in [1, {self.shape.obs_seq_len}])")
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(int(need * 1.0), dtype=torch.uint8, device=dev)
        return self._ws

    # ------------------------------------------------------------------ forward calls
    def _prep(self, state, action, goal, sigma):
        dev = action.device
        if dev.type != "cuda":
            raise RuntimeError("beso_amd: the score-denoising path runs on the GPU only (no CPU fallback); "
                               "move the inputs to the MI355X")
        if state.dim() != 3 or action.dim() != 3:
            raise ValueError("state must be [B,t,obs] and action [B,t,act]")
        B, t, _ = state.shape
        if action.shape[0] != B or action.shape[1] != t or action.shape[2] != self.shape.act_dim:
            raise ValueError(f"action shape {tuple(action.shape)} does not match state {tuple(state.shape)}")
        if state.shape[2] != self.shape.obs_dim:
            raise ValueError(f"state feature dim {state.shape[2]} != obs_dim {self.shape.obs_dim}")
        state = _f32c(state, dev)
        action = _f32c(action, dev)
        G = self.shape.goal_seq_len
        if G > 0:
            if goal is None:
                raise ValueError("goal is required for a goal-conditioned model")
            if goal.dim() == 2:
                goal = goal.unsqueeze(0)
            if goal.shape[0] == 1 and B > 1:
                goal = goal.expand(B, -1, -1)
            if tuple(goal.shape) != (B, G, self.shape.obs_dim):
                raise ValueError(f"goal shape {tuple(goal.shape)} != {(B, G, self.shape.obs_dim)}")
            goal = _f32c(goal, dev)
        else:
            goal = None
        if sigma is not None:
            sigma = _f32c(sigma.reshape(-1), dev)
            if sigma.numel() == 1 and B > 1:
                sigma = sigma.expand(B).contiguous()
            if sigma.numel() != B:
                raise ValueError(f"sigma must have {B} entries, got {sigma.numel()}")
        return dev, B, t, state, action, goal, sigma

    def denoise(self, packed: PackedWeights, state, action, goal, sigma, uncond: bool = False,
                cond_lambda: float = 1.0, precondition: bool = True) -> torch.Tensor:
        """GCDenoiser.forward (precondition=True) or DiffusionGPT.forward (False)."""
        dev, B, t, state, action, goal, sigma = self._prep(state, action, goal, sigma)
        two = precondition and (not uncond) and cond_lambda not in (0.0, 1.0)
        ws = self._workspace(B, t, two, dev)
        out = torch.empty((B, t, self.shape.act_dim), dtype=torch.float32, device=dev)
        gp = goal.data_ptr() if goal is not None else None
        flags = (_lib.FLAG_UNCOND if uncond else 0) | forward_hints()
        with torch.cuda.device(dev):
            if precondition:
                st = self.lib.beso_denoise_fwd(C.byref(self.cfg), packed.buf.data_ptr(), packed.precision,
                                               state.data_ptr(), action.data_ptr(), gp, sigma.data_ptr(),
                                               out.data_ptr(), B, t, flags, float(cond_lambda), ws.data_ptr(),
                                               ws.numel(), _stream_ptr(dev))
            else:
                st = self.lib.beso_score_fwd(C.byref(self.cfg), packed.buf.data_ptr(), packed.precision,
                                             state.data_ptr(), action.data_ptr(), gp, sigma.data_ptr(),
                                             out.data_ptr(), B, t, flags, ws.data_ptr(), ws.numel(),
                                             _stream_ptr(dev))
        _lib.check(st, "denoise_fwd" if precondition else "score_fwd")
        return out

    def sample(self, packed: PackedWeights, sampler: str, state, x_t, goal, sigmas, cond_lambda: float = 1.0,
               inplace: bool = False, stepwise: bool = False) -> torch.Tensor:
        """sample_ddim / sample_euler / sample_heun (s_churn = 0) as ONE enqueue of all steps -- one launch for the whole
        loop where the shape has the one-launch kernel; ``stepwise`` enqueues evaluation by evaluation instead (same
        arithmetic, bit-identical results)."""
        if sampler not in _lib.SAMPLER_IDS:
            raise ValueError("desired sampler type not found!")
        dev, B, t, state, x, goal, _ = self._prep(state, x_t, goal, None)
        if not inplace and x.data_ptr() == x_t.data_ptr():
            x = x.clone()          # the loop updates x in place; keep the caller's x_T intact
        sig = [float(s) for s in (sigmas.detach().cpu().tolist() if torch.is_tensor(sigmas) else sigmas)]
        two = cond_lambda not in (0.0, 1.0)
        ws = self._workspace(B, t, two, dev)
        arr = (C.c_float * len(sig))(*sig)
        gp = goal.data_ptr() if goal is not None else None
        with torch.cuda.device(dev):
            st = self.lib.beso_sample(C.byref(self.cfg), packed.buf.data_ptr(), packed.precision,
                                      _lib.SAMPLER_IDS[sampler], state.data_ptr(), gp, x.data_ptr(), B, t, arr,
                                      len(sig), float(cond_lambda), (_lib.SAMPLE_STEPWISE if stepwise else 0) | forward_hints(),
                                      ws.data_ptr(), ws.numel(), _stream_ptr(dev))
        _lib.check(st, f"sample[{sampler}]")
        return x

    def sample_ancestral(self, packed: PackedWeights, state, x_t, goal, sigmas, cond_lambda: float = 1.0, eta: float = 1.0,
                         noise=None, stepwise: bool = False) -> torch.Tensor:
        """sample_euler_ancestral as ONE enqueue of all steps (``beso_sample_ancestral``) -- one LAUNCH for the whole loop
        where the shape has the one-launch kernel (``stepwise``: evaluation by evaluation, bit-identical).  The per-step noise is drawn
        here, one ``torch.randn_like`` per step that adds noise and in the order of the steps -- the calls the reference's
        loop makes, so a seeded generator gives the same draws as the step-by-step loop; ``noise`` [n_steps, B, t, act]
        injects them instead."""
        dev, B, t, state, x, goal, _ = self._prep(state, x_t, goal, None)
        if x.data_ptr() == x_t.data_ptr():
            x = x.clone()
        sig = [float(s) for s in (sigmas.detach().cpu().tolist() if torch.is_tensor(sigmas) else sigmas)]
        n_steps = len(sig) - 1
        if noise is None:
            noise = torch.zeros((n_steps,) + tuple(x.shape), dtype=torch.float32, device=dev)
            for i in range(n_steps):
                sf, sn = sig[i], sig[i + 1]
                up = min(sn, eta * (sn ** 2 * (sf ** 2 - sn ** 2) / sf ** 2) ** 0.5) if eta else 0.0
                if sn ** 2 - up ** 2 > 0:                # sigma_down > 0: the step draws (gc_sampling.py:246-247)
                    noise[i] = torch.randn_like(x)
        else:
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            if noise.shape != (n_steps,) + tuple(x.shape):
                raise ValueError("noise must be [len(sigmas) - 1, B, t, act]")
        two = cond_lambda not in (0.0, 1.0)
        ws = self._workspace(B, t, two, dev)
        arr = (C.c_float * len(sig))(*sig)
        gp = goal.data_ptr() if goal is not None else None
        with torch.cuda.device(dev):
            st = self.lib.beso_sample_ancestral(C.byref(self.cfg), packed.buf.data_ptr(), packed.precision, state.data_ptr(), gp,
                                                x.data_ptr(), B, t, arr, len(sig), float(cond_lambda), float(eta),
                                                noise.data_ptr(), (_lib.SAMPLE_STEPWISE if stepwise else 0) | forward_hints(),
                                                ws.data_ptr(), ws.numel(), _stream_ptr(dev))
        _lib.check(st, "sample[euler_ancestral]")
        return x

    # ------------------------------------------------------------------ profiling hooks (bench.py)
    def profile_enable(self, site: str) -> None:
        self.lib.beso_profile_enable(_lib.SITES[site])

    def profile_read(self):
        ms, n = C.c_double(0.0), C.c_int(0)
        _lib.check(self.lib.beso_profile_read(C.byref(ms), C.byref(n)), "profile_read")
        return ms.value, n.value
