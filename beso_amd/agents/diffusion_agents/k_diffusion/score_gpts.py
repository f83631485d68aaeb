"""Drop-in for ``beso.agents.diffusion_agents.k_diffusion.score_gpts.DiffusionGPT`` whose forward runs
as hand-written HIP kernels on MI355X (libbeso_hip.so, include/beso_hip.h).

Same constructor kwargs, same parameter/buffer names and order (so reference checkpoints load and
the EMA helper's zip over ``parameters()`` stays aligned), same call signature
``model(states, actions, goals, sigma, uncond=False, keep_last_actions=False)``
(reference: score_gpts.py:121-139, 272-358).

The module is a parameter container.  Inference (no autograd) never touches torch ops: it packs the
parameters into the kernel image (cached, re-packed when a parameter changes) and calls
``beso_score_fwd``.  Only when autograd is required (``train_step``) does it evaluate the same
function with differentiable torch ops on the GPU -- the HIP backward is a later row of the scope
table.  There is no CPU path.
"""
from __future__ import annotations

import contextlib
import math
import os
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ....runtime import PackedWeights, ScoreNetRuntime, ScoreNetShape

DEFAULT_PRECISION = os.environ.get("BESO_AMD_PRECISION", "bf16")


class _SelfAttentionParams(nn.Module):
    """key/query/value/proj Linear layers + the lower-triangular ``mask`` buffer, named as in the
    reference so ``state_dict`` keys match (score_gpts.py:33-47)."""

    def __init__(self, n_embd, n_heads, attn_pdrop, resid_pdrop, block_size):
        super().__init__()
        if n_embd % n_heads:
            raise AssertionError("n_embd must be divisible by n_heads")
        self.key = nn.Linear(n_embd, n_embd)
        self.query = nn.Linear(n_embd, n_embd)
        self.value = nn.Linear(n_embd, n_embd)
        self.attn_drop = nn.Dropout(attn_pdrop)
        self.resid_drop = nn.Dropout(resid_pdrop)
        self.proj = nn.Linear(n_embd, n_embd)
        self.register_buffer("mask", torch.ones(block_size, block_size).tril_().view(1, 1, block_size, block_size))
        self.n_head = n_heads


class _BlockParams(nn.Module):
    def __init__(self, n_embd, n_heads, attn_pdrop, resid_pdrop, block_size):
        super().__init__()
        self.ln1 = nn.LayerNorm(n_embd)
        self.ln2 = nn.LayerNorm(n_embd)
        self.attn = _SelfAttentionParams(n_embd, n_heads, attn_pdrop, resid_pdrop, block_size)
        self.mlp = nn.Sequential(nn.Linear(n_embd, 4 * n_embd), nn.GELU(), nn.Linear(4 * n_embd, n_embd),
                                 nn.Dropout(resid_pdrop))


class DiffusionGPT(nn.Module):
    """Score transformer over ``[sigma, g_1..g_G, s_1, a_1, ..., s_t, a_t]``."""

    def __init__(self, state_dim: int, device: str, goal_conditioned: bool, action_dim: int, embed_dim: int,
                 embed_pdrob: float, attn_pdrop: float, resid_pdrop: float, n_layers: int, n_heads: int,
                 goal_seq_len: int, obs_seq_len: int, sigma_vocab_size: int = None, time_embedding_fn=None,
                 goal_drop: float = 0, linear_output=False, precision: Optional[str] = None):
        super().__init__()
        # sigma_vocab_size / time_embedding_fn are accepted and ignored, as in the reference (:136,178)
        self.device = device
        self.goal_conditioned = goal_conditioned
        if not goal_conditioned:
            goal_seq_len = 0                                   # :143-144
        self.block_size = goal_seq_len + 2 * obs_seq_len + 1   # :148
        seq_size = goal_seq_len + obs_seq_len + 1              # :150
        self.tok_emb = nn.Linear(state_dim, embed_dim)
        self.pos_emb = nn.Parameter(torch.zeros(1, seq_size, embed_dim))
        self.drop = nn.Dropout(embed_pdrob)
        self.cond_mask_prob = goal_drop
        self.action_dim = action_dim
        self.obs_dim = state_dim
        self.embed_dim = embed_dim
        self.blocks = nn.Sequential(*[_BlockParams(embed_dim, n_heads, attn_pdrop, resid_pdrop, self.block_size)
                                      for _ in range(n_layers)])
        self.ln_f = nn.LayerNorm(embed_dim)
        self.goal_seq_len = goal_seq_len
        self.obs_seq_len = obs_seq_len
        self.sigma_emb = nn.Linear(1, embed_dim)
        self.action_emb = nn.Linear(action_dim, embed_dim)
        if linear_output:
            self.action_pred = nn.Linear(embed_dim, action_dim)
        else:
            self.action_pred = nn.Sequential(nn.Linear(embed_dim, 100), nn.SiLU(), nn.Linear(100, action_dim))
        self.linear_output = bool(linear_output)
        self.n_heads = n_heads
        self.n_layers = n_layers
        self._pdrops = (float(embed_pdrob), float(attn_pdrop), float(resid_pdrop))
        self._reset_parameters()
        # --- HIP runtime state (not part of state_dict) ---
        self.precision = precision or DEFAULT_PRECISION
        self._runtimes = {}
        self._packed: Optional[PackedWeights] = None
        self._override: Optional[PackedWeights] = None
        self._dirty_epoch = 0

    # ------------------------------------------------------------------ init / bookkeeping
    def _reset_parameters(self):
        """N(0, 0.02) Linear weights, zero biases, unit LayerNorm, N(0, 0.02) pos_emb (:202-211)."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0.0, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        nn.init.normal_(self.pos_emb, mean=0.0, std=0.02)

    def get_block_size(self):
        return self.block_size

    def get_params(self):
        return self.parameters()

    def shape(self, sigma_data: float = 1.0) -> ScoreNetShape:
        return ScoreNetShape(self.obs_dim, self.action_dim, self.embed_dim, self.n_layers, self.n_heads,
                             self.goal_seq_len, self.obs_seq_len, self.linear_output, float(sigma_data))

    def runtime(self, sigma_data: float = 1.0) -> ScoreNetRuntime:
        key = (float(sigma_data), self.precision)
        rt = self._runtimes.get(key)
        if rt is None:
            rt = self._runtimes[key] = ScoreNetRuntime(self.shape(sigma_data), self.precision)
        return rt

    def set_precision(self, precision: str) -> None:
        """'bf16' (throughput), 'fp32' (parity: exact-fp32 MFMA)."""
        if precision != self.precision:
            self.precision = precision
            self._packed = None

    def mark_weights_dirty(self) -> None:
        """Call after modifying parameters through ``.data`` (which bypasses version counters)."""
        self._dirty_epoch += 1

    def _weights_key(self):
        return (self._dirty_epoch, self.precision) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def packed_weights(self) -> PackedWeights:
        """Kernel image of the CURRENT parameter values (or the image installed by ``use_weights``)."""
        if self._override is not None:
            return self._override
        key = self._weights_key()
        if self._packed is None or self._packed.key != key:
            self._packed = self.runtime().pack(list(self.parameters()), key=key, into=self._packed)
        return self._packed

    def pack_external(self, tensors, into: Optional[PackedWeights] = None) -> PackedWeights:
        """Kernel image of another set of values for the same parameters (e.g. the EMA shadow)."""
        return self.runtime().pack(list(tensors), key=None, into=into)

    @contextlib.contextmanager
    def use_weights(self, packed: Optional[PackedWeights]):
        """Evaluate with ``packed`` instead of the live parameters (EMA inference without the
        store / copy_to / restore round trip of beso_agent.py:343-381)."""
        prev, self._override = self._override, packed
        try:
            yield self
        finally:
            self._override = prev

    # ------------------------------------------------------------------ forward
    def _hip_eligible(self, *tensors) -> bool:
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                        or any(torch.is_tensor(t) and t.requires_grad for t in tensors)):
            return False
        if self.training and any(p > 0 for p in self._pdrops):
            return False        # dropout is part of the training-mode function
        return True

    def forward(self, states, actions, goals, sigma, uncond: Optional[bool] = False,
                keep_last_actions: Optional[bool] = False):
        b, t, _ = states.size()
        assert t <= self.block_size, "Cannot forward, model block size is exhausted."
        if self.training:
            goals = self.mask_cond(goals)                                   # :298-299
        if self._hip_eligible(states, actions, goals, sigma):
            pred = self.runtime().denoise(self.packed_weights(), states, actions, goals, sigma,
                                          uncond=bool(uncond), precondition=False)
        else:
            pred = self._forward_autograd(states, actions, goals, sigma, bool(uncond))
        if keep_last_actions:                                               # :355-356
            pred = torch.cat([actions[:, :-1, :], pred[:, -1, :].reshape(1, 1, -1)], dim=1)
        return pred

    def mask_cond(self, cond, force_mask=False):
        """Training-time goal dropout for classifier-free guidance: elementwise Bernoulli (:360-371)."""
        if force_mask:
            return torch.zeros_like(cond)
        if self.training and self.cond_mask_prob > 0.:
            keep = 1. - torch.bernoulli(torch.full_like(cond, self.cond_mask_prob))
            return cond * keep
        return cond

    # differentiable evaluation for the training step (torch ops on the GPU; autograd supplies the
    # backward).  Semantics identical to the HIP forward: tests/test_gpu_parity.py compares them.
    def _forward_autograd(self, states, actions, goals, sigma, uncond: bool):
        b, t, _ = states.shape
        G, D, H = self.goal_seq_len, self.embed_dim, self.n_heads
        emb_t = self.sigma_emb((sigma.log() / 4).reshape(b, 1).to(torch.float32)).unsqueeze(1)
        pos = self.pos_emb[:, : t + G, :]
        s_x = self.drop(self.tok_emb(states) + pos[:, G:, :])
        a_x = self.drop(self.action_emb(actions) + pos[:, G:, :])
        seq = [emb_t]
        if self.goal_conditioned:
            if uncond:
                goals = torch.zeros_like(goals)
            seq.append(self.drop(self.tok_emb(goals) + pos[:, :G, :]).expand(b, -1, -1))
        seq.append(torch.stack((s_x, a_x), dim=2).reshape(b, 2 * t, D))       # s_1,a_1,s_2,a_2,...
        x = torch.cat(seq, dim=1)
        T = x.shape[1]
        for blk in self.blocks:
            at = blk.attn
            h = blk.ln1(x)
            q, k, v = (lin(h).view(b, T, H, D // H).transpose(1, 2) for lin in (at.query, at.key, at.value))
            w = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(D // H))
            w = w.masked_fill(at.mask[:, :, :T, :T] == 0, float("-inf"))
            w = at.attn_drop(F.softmax(w, dim=-1))
            y = (w @ v).transpose(1, 2).reshape(b, T, D)
            x = x + at.resid_drop(at.proj(y))
            x = x + blk.mlp(blk.ln2(x))
        x = self.ln_f(x)[:, G + 1:, :]
        a_out = x.reshape(b, x.shape[1] // 2, 2, D)[:, :, 1, :]
        return self.action_pred(a_out)
