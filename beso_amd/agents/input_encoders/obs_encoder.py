"""Boundary plumbing of the agent: which entries of a batch dict are the observation window and the goal,
and on which device the score network wants them (interface of the reference's
beso/agents/input_encoders/obs_encoder.py:11-21: constructor kwargs ``device, state_modality,
goal_modality``; ``forward(batch) -> (state, goal-or-None)``)."""
from typing import Optional, Tuple

import torch
from torch import Tensor, nn


class NoEncoder(nn.Module):
    """Identity "encoder": observations are already feature vectors (kitchen / block-push)."""

    def __init__(self, device: str, state_modality: str, goal_modality: str):
        super().__init__()
        self.device = device
        self.state_modality, self.goal_modality = state_modality, goal_modality

    def _fetch(self, batch: dict, key: str) -> Optional[Tensor]:
        value = batch.get(key)
        return None if value is None else value.to(self.device, non_blocking=True)

    @torch.no_grad()
    def forward(self, batch: dict) -> Tuple[Tensor, Optional[Tensor]]:
        state = self._fetch(batch, self.state_modality)
        if state is None:
            raise KeyError(self.state_modality)
        return state, self._fetch(batch, self.goal_modality)
