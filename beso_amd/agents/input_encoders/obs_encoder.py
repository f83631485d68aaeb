"""Boundary plumbing: picks the state / goal tensors out of a batch dict and moves them to the
device (reference: beso/agents/input_encoders/obs_encoder.py:11-21)."""
import torch
import torch.nn as nn


class NoEncoder(nn.Module):
    def __init__(self, device: str, state_modality: str, goal_modality: str):
        super().__init__()
        self.state_modality = state_modality
        self.goal_modality = goal_modality
        self.device = device

    @torch.no_grad()
    def forward(self, x: dict):
        state = x[self.state_modality].to(self.device)
        goal = x[self.goal_modality].to(self.device) if self.goal_modality in x else None
        return state, goal
