"""Synthetic workloads for measurement: the shipped model shapes, seeded weights and seeded inputs.

The reference ships no checkpoints (``.MISSING_LARGE_BLOBS``) and the build box has no data sets, so ``bench.py`` and the
tools under ``tools/`` time the kernels on seeded tensors of the shipped shapes:

* shapes: kitchen ``configs/franka_kitchen_main_config.yaml:26-40,58,61``, block-push
  ``configs/block_push_main_config.yaml:27-42,59``, the long-horizon variant of BASELINE config 5;
* weights: the distribution of the reference's init (``score_gpts.py:202-211``: N(0, std) for Linear weights and
  ``pos_emb``, LayerNorm weight 1) with biases and LayerNorm affine parameters perturbed as well so that no term of the
  forward is multiplied by zero; one PCG64 stream per tensor, seeded with (seed, tensor index);
* inputs: N(0, 1) states / goals / x_T (``beso_agent.py:274``), one PCG64 stream per call.

The same recipe is written down a second time in ``oracle/beso_oracle.py`` (the test oracle keeps no dependency on this
package); ``tests/test_host_logic.py`` checks that the two agree.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

HEAD_HIDDEN = 100        # width of the MLP action head (score_gpts.py:187)


@dataclass(frozen=True)
class ModelShape:
    obs_dim: int
    act_dim: int
    embed_dim: int
    n_layers: int
    n_heads: int
    goal_seq_len: int
    obs_seq_len: int
    goal_conditioned: bool = True
    linear_output: bool = True
    sigma_data: float = 0.5

    @property
    def G(self) -> int:                                   # goals are dropped when not goal conditioned (score_gpts.py:143-144)
        return self.goal_seq_len if self.goal_conditioned else 0

    @property
    def block_size(self) -> int:                          # tokens of a full window (score_gpts.py:148)
        return self.G + 2 * self.obs_seq_len + 1

    @property
    def seq_size(self) -> int:                            # rows of pos_emb (score_gpts.py:150)
        return self.G + self.obs_seq_len + 1

    def flops_per_sample(self, t: Optional[int] = None) -> int:
        """Algorithmic FLOPs of one score-net forward of one sample (multiply-add = 2, full T x T attention)."""
        t = self.obs_seq_len if t is None else t
        D, T = self.embed_dim, 1 + self.G + 2 * t
        per_layer = 24 * T * D * D + 4 * T * T * D
        embed = 2 * D * (t * self.obs_dim + self.G * self.obs_dim + t * self.act_dim + 1)
        return self.n_layers * per_layer + embed + 2 * t * D * self.act_dim

    def as_dict(self) -> dict:
        return asdict(self)


SHAPES: Dict[str, ModelShape] = {
    "kitchen": ModelShape(30, 9, 360, 6, 6, 2, 4),
    "block_push": ModelShape(10, 2, 240, 4, 12, 1, 5),
    "long_horizon": ModelShape(30, 9, 512, 6, 8, 2, 32),
    "tiny": ModelShape(7, 3, 48, 2, 6, 2, 3),
    "tiny_mlp_head": ModelShape(5, 2, 32, 1, 4, 1, 2, linear_output=False, sigma_data=1.0),
    "tiny_nogoal": ModelShape(6, 4, 40, 2, 5, 2, 3, goal_conditioned=False),
}


def parameter_table(shape: ModelShape) -> List[Tuple[str, tuple]]:
    """(state_dict key, tensor shape) of ``GCDenoiser(DiffusionGPT)`` in ``named_parameters()`` order."""
    D, root = shape.embed_dim, "inner_model."
    rows = [(root + "pos_emb", (1, shape.seq_size, D)), (root + "tok_emb.weight", (D, shape.obs_dim)), (root + "tok_emb.bias", (D,))]
    for i in range(shape.n_layers):
        blk = f"{root}blocks.{i}."
        for ln in ("ln1", "ln2"):
            rows += [(f"{blk}{ln}.weight", (D,)), (f"{blk}{ln}.bias", (D,))]
        for lin in ("key", "query", "value", "proj"):
            rows += [(f"{blk}attn.{lin}.weight", (D, D)), (f"{blk}attn.{lin}.bias", (D,))]
        rows += [(blk + "mlp.0.weight", (4 * D, D)), (blk + "mlp.0.bias", (4 * D,)), (blk + "mlp.2.weight", (D, 4 * D)),
                 (blk + "mlp.2.bias", (D,))]
    rows += [(root + "ln_f.weight", (D,)), (root + "ln_f.bias", (D,)), (root + "sigma_emb.weight", (D, 1)),
             (root + "sigma_emb.bias", (D,)), (root + "action_emb.weight", (D, shape.act_dim)), (root + "action_emb.bias", (D,))]
    if shape.linear_output:
        rows += [(root + "action_pred.weight", (shape.act_dim, D)), (root + "action_pred.bias", (shape.act_dim,))]
    else:
        rows += [(root + "action_pred.0.weight", (HEAD_HIDDEN, D)), (root + "action_pred.0.bias", (HEAD_HIDDEN,)),
                 (root + "action_pred.2.weight", (shape.act_dim, HEAD_HIDDEN)), (root + "action_pred.2.bias", (shape.act_dim,))]
    return rows


def make_weights(shape: ModelShape, seed: int = 0, std: float = 0.02, bias_std: Optional[float] = None) -> Dict[str, np.ndarray]:
    bias_std = std if bias_std is None else bias_std
    out = {}
    for index, (key, dims) in enumerate(parameter_table(shape)):
        draw = np.random.Generator(np.random.PCG64([seed, index])).standard_normal(dims, dtype=np.float32)
        is_ln = ".ln" in key or "ln_f" in key
        if is_ln and key.endswith("weight"):
            val = 1.0 + 0.1 * draw
        elif is_ln:
            val = 0.1 * draw
        elif key.endswith("bias"):
            val = bias_std * draw
        else:
            val = std * draw
        out[key] = np.ascontiguousarray(val, dtype=np.float32)
    return out


def make_inputs(shape: ModelShape, batch: int, seed: int = 0, t: Optional[int] = None, sigma_max: float = 1.0):
    """(state [B,t,obs], goal [B,max(G,1),obs], x_T [B,t,act]) from one seeded stream."""
    t = shape.obs_seq_len if t is None else t
    gen = np.random.Generator(np.random.PCG64([seed, 9001]))
    state = gen.standard_normal((batch, t, shape.obs_dim), dtype=np.float32)
    goal = gen.standard_normal((batch, max(shape.goal_seq_len, 1), shape.obs_dim), dtype=np.float32)
    x_t = gen.standard_normal((batch, t, shape.act_dim), dtype=np.float32) * np.float32(sigma_max)
    return state, goal, x_t
