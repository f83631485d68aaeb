"""Differentiable torch-op evaluation of the score network and of the score-matching loss over the parameters of a
``beso_amd`` module -- TEST INFRASTRUCTURE (the comparator of the HIP training step, and the thing the CPU suite pins
to the reference's ``loss.backward()`` fixtures).  The product has no torch-op evaluation of the network: a call the
HIP kernels cannot serve raises there.

Restates score_gpts.py:272-358 (DiffusionGPT.forward), :50-115 (blocks), score_wrappers.py:45-79 (GCDenoiser.loss)
of the reference over the drop-in module's own nn.Parameters (same names, same order)."""
import math

import torch
import torch.nn.functional as F


def forward_autograd(inner, states, actions, goals, sigma, uncond: bool = False):
    """DiffusionGPT.forward with torch ops; training mode applies the module's nn.Dropout layers and mask_cond."""
    b, t, _ = states.shape
    G, D, H = inner.goal_seq_len, inner.embed_dim, inner.n_heads
    if inner.training and goals is not None:
        goals = inner.mask_cond(goals)                                     # score_gpts.py:298-299
    emb_t = inner.sigma_emb((sigma.log() / 4).reshape(b, 1).to(torch.float32)).unsqueeze(1)
    pos = inner.pos_emb[:, : t + G, :]
    s_x = inner.drop(inner.tok_emb(states) + pos[:, G:, :])
    a_x = inner.drop(inner.action_emb(actions) + pos[:, G:, :])
    seq = [emb_t]
    if inner.goal_conditioned:
        if uncond:
            goals = torch.zeros_like(goals)
        seq.append(inner.drop(inner.tok_emb(goals) + pos[:, :G, :]).expand(b, -1, -1))
    seq.append(torch.stack((s_x, a_x), dim=2).reshape(b, 2 * t, D))           # s_1,a_1,s_2,a_2,...
    x = torch.cat(seq, dim=1)
    T = x.shape[1]
    for blk in inner.blocks:
        at = blk.attn
        h = blk.ln1(x)
        q, k, v = (lin(h).view(b, T, H, D // H).transpose(1, 2) for lin in (at.query, at.key, at.value))
        w = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(D // H))
        w = w.masked_fill(at.mask[:, :, :T, :T] == 0, float("-inf"))
        w = at.attn_drop(F.softmax(w, dim=-1))
        y = (w @ v).transpose(1, 2).reshape(b, T, D)
        x = x + at.resid_drop(at.proj(y))
        x = x + blk.mlp(blk.ln2(x))
    x = inner.ln_f(x)[:, G + 1:, :]
    a_out = x.reshape(b, x.shape[1] // 2, 2, D)[:, :, 1, :]
    return inner.action_pred(a_out)


def loss_autograd(model, state, action, goal, noise, sigma, pred_last_action_only: bool = False):
    """GCDenoiser.loss (score_wrappers.py:45-79) over forward_autograd; mutates ``noise`` like the reference (:63)."""
    if pred_last_action_only:
        noise[:, :-1, :] = 0
    sd2 = model.sigma_data ** 2
    sig = sigma.reshape(-1, *([1] * (action.ndim - 1)))
    total = sig ** 2 + sd2
    c_skip, c_out, c_in = sd2 / total, sig * model.sigma_data / total ** 0.5, 1 / total ** 0.5
    noised = action + noise * sig
    out = forward_autograd(model.inner_model, state, noised * c_in, goal, sigma)
    target = (action - c_skip * noised) / c_out
    if pred_last_action_only:
        return (out[:, -1, :] - target[:, -1, :]).pow(2).mean()
    return (out - target).pow(2).flatten(1).mean()


def install_autograd_training():
    """The same patch without pytest's monkeypatch (spawned worker processes): returns a function that undoes it."""
    from beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser
    saved = GCDenoiser.loss, GCDenoiser.hip_train_step

    def loss(self, state, action, goal, noise, sigma, **kwargs):
        return loss_autograd(self, state, action, goal, noise, sigma, bool(kwargs.get("pred_last_action_only", False)))

    GCDenoiser.loss = loss
    GCDenoiser.hip_train_step = lambda self, *a, **k: None

    def undo():
        GCDenoiser.loss, GCDenoiser.hip_train_step = saved
    return undo


def use_autograd_training(monkeypatch):
    """Make GCDenoiser.loss / BesoAgent.train_step of THIS test run on the torch-autograd comparator instead of the HIP
    step (monkeypatch scope)."""
    from beso_amd.agents.diffusion_agents.k_diffusion.score_wrappers import GCDenoiser

    def loss(self, state, action, goal, noise, sigma, **kwargs):
        return loss_autograd(self, state, action, goal, noise, sigma, bool(kwargs.get("pred_last_action_only", False)))

    monkeypatch.setattr(GCDenoiser, "loss", loss)
    monkeypatch.setattr(GCDenoiser, "hip_train_step", lambda self, *a, **k: None)
