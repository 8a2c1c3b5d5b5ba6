"""Multi-scale trajectory critics: spectral-normalised MLPs over flattened [B, T*2] trajectories.

State_dict layout and behaviour follow ``models/critics/base_critic.py:13-44`` (hook-based
``torch.nn.utils.spectral_norm``: one power iteration per training-mode forward; keys
``mlp.{0,3,6,9}.{weight_orig,weight_u,weight_v,bias}``, ``out.*``) and the three horizon
wrappers (``short_scale_critic.py``, ``mid_scale_critic.py``, ``long_scale_critic.py``).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm


class BaseTrajectoryCritic(nn.Module):

    def __init__(self, input_dim: int = 2, hidden_dim: int = 128, num_layers: int = 3, dropout: float = 0.1) -> None:
        super().__init__()
        blocks, width = [], input_dim
        for _ in range(num_layers):
            blocks += [spectral_norm(nn.Linear(width, hidden_dim)), nn.LeakyReLU(0.2, inplace=True), nn.Dropout(dropout)]
            width = hidden_dim
        self.mlp = nn.Sequential(*blocks)
        self.out = spectral_norm(nn.Linear(hidden_dim, 1))

    def forward(self, traj: torch.Tensor) -> torch.Tensor:
        """traj [B, T, D] -> realism score [B]."""
        return self.out(self.mlp(traj.reshape(traj.shape[0], -1))).squeeze(-1)


class _HorizonCritic(BaseTrajectoryCritic):
    _layers = 3

    def __init__(self, horizon: int, input_dim: int = 2, hidden_dim: int = 256) -> None:
        self.horizon = horizon
        super().__init__(input_dim=horizon * input_dim, hidden_dim=hidden_dim, num_layers=self._layers)

    def forward(self, traj: torch.Tensor) -> torch.Tensor:
        return super().forward(traj[:, :self.horizon])


class ShortScaleCritic(_HorizonCritic):
    """First ``horizon`` steps, 2 hidden layers (``short_scale_critic.py:11-29``)."""
    _layers = 2

    def __init__(self, horizon: int = 10, input_dim: int = 2, hidden_dim: int = 128) -> None:
        super().__init__(horizon, input_dim, hidden_dim)


class MidScaleCritic(_HorizonCritic):
    """First ``horizon`` steps, 3 hidden layers (``mid_scale_critic.py:11-26``)."""
    _layers = 3

    def __init__(self, horizon: int = 20, input_dim: int = 2, hidden_dim: int = 256) -> None:
        super().__init__(horizon, input_dim, hidden_dim)


class LongScaleCritic(_HorizonCritic):
    """Full horizon, 4 hidden layers (``long_scale_critic.py:11-26``)."""
    _layers = 4

    def __init__(self, horizon: int, input_dim: int = 2, hidden_dim: int = 256) -> None:
        super().__init__(horizon, input_dim, hidden_dim)
