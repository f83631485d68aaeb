"""CPU oracle, ATen backend  --  TEST INFRASTRUCTURE ONLY (same rules as beso_oracle.py: only ``tests/`` and the
``cpu_baseline`` leg of ``bench.py`` import it; nothing under ``beso_amd/`` does).

The same restatement of GCDenoiser -> DiffusionGPT -> sample_ddim as ``beso_oracle.py``, written over the operators the
REFERENCE's CPU path executes (``F.linear`` = addmm, ``F.layer_norm``, exact-erf ``F.gelu``, ``softmax``, batched
``matmul``; fp32, ``torch.set_num_threads`` host threads) instead of numpy: this is what ``bench.py`` times as the CPU
baseline, so that the baseline costs what the reference costs on the same cores (addmm 50 %,
gelu 30 %, copies 20 % of the reference's CPU time) -- the numpy port spends most of its time in scipy's erf and in
temporaries the reference never makes.  It is pinned to the same reference-generated vectors (tests/test_oracle_golden.py).

Citations are file:line relative to the reference checkout (intuitive-robots/beso).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

P = "inner_model."


def to_torch(w: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    """The oracle's flat weight dict (state_dict keys, torch Linear layout [out, in]) as CPU fp32 tensors."""
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in w.items()}


@torch.no_grad()
def score_gpt_forward(W: Dict[str, torch.Tensor], cfg, states, actions, goals, sigma, uncond: bool = False):
    """DiffusionGPT.forward, eval mode (score_gpts.py:272-358)."""
    b, t, _ = states.shape
    D, H, G = cfg.embed_dim, cfg.n_heads, cfg.G
    assert t <= cfg.block_size                                                               # :282
    emb_t = F.linear((sigma.log() / 4).reshape(b, 1).to(torch.float32), W[P + "sigma_emb.weight"],
                     W[P + "sigma_emb.bias"]).reshape(b, 1, D)                               # :284-288
    pos = W[P + "pos_emb"][:, : t + G, :]                                                    # :311-318
    state_x = F.linear(states, W[P + "tok_emb.weight"], W[P + "tok_emb.bias"]) + pos[:, G:, :]          # :305,323
    action_x = F.linear(actions, W[P + "action_emb.weight"], W[P + "action_emb.bias"]) + pos[:, G:, :]  # :307,325
    sa = torch.stack([state_x, action_x], dim=2).reshape(b, 2 * t, D)                        # :330-331
    if cfg.goal_conditioned:
        if goals.dim() == 2:
            goals = goals.unsqueeze(0).expand(b, -1, -1)
        if uncond:
            goals = torch.zeros_like(goals)                                                  # :301-302
        goal_x = F.linear(goals, W[P + "tok_emb.weight"], W[P + "tok_emb.bias"]) + pos[:, :G, :]        # :306,322
        x = torch.cat([emb_t, goal_x, sa], dim=1)                                            # :335
    else:
        x = torch.cat([emb_t, sa], dim=1)                                                    # :337
    T = x.shape[1]
    hd = D // H
    causal = torch.tril(torch.ones(T, T, dtype=torch.bool))                                  # :42-47
    for i in range(cfg.n_layers):                                                            # Block.forward :112-115
        B_ = f"{P}blocks.{i}."
        h = F.layer_norm(x, (D,), W[B_ + "ln1.weight"], W[B_ + "ln1.bias"])
        k = F.linear(h, W[B_ + "attn.key.weight"], W[B_ + "attn.key.bias"]).view(b, T, H, hd).transpose(1, 2)      # :58-66
        q = F.linear(h, W[B_ + "attn.query.weight"], W[B_ + "attn.query.bias"]).view(b, T, H, hd).transpose(1, 2)
        v = F.linear(h, W[B_ + "attn.value.weight"], W[B_ + "attn.value.bias"]).view(b, T, H, hd).transpose(1, 2)
        att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd))                              # :69
        att = att.masked_fill(~causal, float("-inf"))                                        # :70
        att = F.softmax(att, dim=-1)                                                         # :71
        y = (att @ v).transpose(1, 2).contiguous().view(b, T, D)                             # :73-76
        x = x + F.linear(y, W[B_ + "attn.proj.weight"], W[B_ + "attn.proj.bias"])            # :79,113
        h = F.layer_norm(x, (D,), W[B_ + "ln2.weight"], W[B_ + "ln2.bias"])
        h = F.gelu(F.linear(h, W[B_ + "mlp.0.weight"], W[B_ + "mlp.0.bias"]))                # :105-108 (erf form)
        x = x + F.linear(h, W[B_ + "mlp.2.weight"], W[B_ + "mlp.2.bias"])                    # :114
    x = F.layer_norm(x, (D,), W[P + "ln_f.weight"], W[P + "ln_f.bias"])                      # :341
    x = x[:, (G + 1):, :]                                                                    # :344
    a_out = x.reshape(b, x.shape[1] // 2, 2, D)[:, :, 1, :]                                  # :347-353
    if cfg.linear_output:
        return F.linear(a_out, W[P + "action_pred.weight"], W[P + "action_pred.bias"])       # :354
    return F.linear(F.silu(F.linear(a_out, W[P + "action_pred.0.weight"], W[P + "action_pred.0.bias"])),
                    W[P + "action_pred.2.weight"], W[P + "action_pred.2.bias"])


@torch.no_grad()
def denoise(W, cfg, state, action, goal, sigma, uncond: bool = False):
    """GCDenoiser.forward (score_wrappers.py:31-43, 95-96)."""
    sd = cfg.sigma_data
    s = sigma.reshape(-1, 1, 1)
    c_skip = sd ** 2 / (s ** 2 + sd ** 2)
    c_out = s * sd / (s ** 2 + sd ** 2) ** 0.5
    c_in = 1 / (s ** 2 + sd ** 2) ** 0.5
    return score_gpt_forward(W, cfg, state, action * c_in, goal, sigma, uncond=uncond) * c_out + action * c_skip


@torch.no_grad()
def sample_ddim(W, cfg, state, action, goal, sigmas):
    """sample_ddim (gc_sampling.py:895-924) over ``denoise``."""
    s_in = action.new_ones([action.shape[0]])
    for i in range(len(sigmas) - 1):
        denoised = denoise(W, cfg, state, action, goal, sigmas[i] * s_in)                    # :918
        t, t_next = -sigmas[i].log(), -sigmas[i + 1].log()                                   # :921
        h = t_next - t
        action = ((-t_next).exp() / (-t).exp()) * action - (-h).expm1() * denoised           # :922-923
    return action
