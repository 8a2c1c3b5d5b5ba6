"""Multi-modal trajectory decoder (dense; stock GEMMs).  Parameter names follow
``models/decoder.py:74-139`` (``aggr_embed.{0,1}``, ``loc.{0,1,3}``, ``scale.{0,1,3}``, ``pi.{0,1,3,4,6}``)."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .embedding import init_weights


def _head(width: int, out: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(width, width), nn.LayerNorm(width), nn.ReLU(inplace=True), nn.Linear(width, out))


class MLPDecoder(nn.Module):

    def __init__(self, local_channels: int, global_channels: int, future_steps: int, num_modes: int,
                 uncertain: bool = True, min_scale: float = 1e-3) -> None:
        super().__init__()
        self.input_size = global_channels
        self.hidden_size = local_channels
        self.future_steps = future_steps
        self.num_modes = num_modes
        self.uncertain = uncertain
        self.min_scale = min_scale
        hid = self.hidden_size
        self.aggr_embed = nn.Sequential(nn.Linear(self.input_size + hid, hid), nn.LayerNorm(hid), nn.ReLU(inplace=True))
        self.loc = _head(hid, future_steps * 2)
        if uncertain:
            self.scale = _head(hid, future_steps * 2)
        self.pi = nn.Sequential(nn.Linear(hid + self.input_size, hid), nn.LayerNorm(hid), nn.ReLU(inplace=True),
                                nn.Linear(hid, hid), nn.LayerNorm(hid), nn.ReLU(inplace=True), nn.Linear(hid, 1))
        self.apply(init_weights)

    def forward(self, local_embed: torch.Tensor, global_embed: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """local_embed [N, D], global_embed [F, N, D] -> (y_hat [F, N, H, 4 or 2], pi [N, F])."""
        tiled = local_embed.expand(self.num_modes, *local_embed.shape)
        pi = self.pi(torch.cat((tiled, global_embed), dim=-1)).squeeze(-1).t()
        out = self.aggr_embed(torch.cat((global_embed, tiled), dim=-1))
        loc = self.loc(out).view(self.num_modes, -1, self.future_steps, 2)
        if not self.uncertain:
            return loc, pi
        scale = F.elu(self.scale(out), alpha=1.0) + 1.0
        scale = torch.clamp(scale.view(self.num_modes, -1, self.future_steps, 2), min=0.05)
        return torch.cat((loc, scale), dim=-1), pi
