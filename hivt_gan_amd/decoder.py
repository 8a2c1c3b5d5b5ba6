"""Multi-modal trajectory decoder.  Parameter names follow ``models/decoder.py:74-139``
(``aggr_embed.{0,1}``, ``loc.{0,1,3}``, ``scale.{0,1,3}``, ``pi.{0,1,3,4,6}``).  On the GPU every
Linear (+LayerNorm +ReLU) group is one fused launch; CPU tensors run the stock layers."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .embedding import init_weights


def _run(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """``seq(x)`` for a stack of Linear [, LayerNorm [, ReLU]] groups, one fused launch per group on the GPU."""
    if not x.is_cuda:
        return seq(x)
    mods = list(seq)
    i = 0
    while i < len(mods):
        lin = mods[i]
        norm = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.LayerNorm) else None
        j = i + (2 if norm is not None else 1)
        relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
        x = ops.fused_linear(x, lin.weight, lin.bias, layer_norm=norm, relu=relu)
        i = j + (1 if relu else 0)
    return x


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
        pi = _run(self.pi, torch.cat((tiled, global_embed), dim=-1)).squeeze(-1).t()
        out = _run(self.aggr_embed, torch.cat((global_embed, tiled), dim=-1))
        loc = _run(self.loc, out).view(self.num_modes, -1, self.future_steps, 2)
        if not self.uncertain:
            return loc, pi
        scale = F.elu(_run(self.scale, out), alpha=1.0) + 1.0
        scale = torch.clamp(scale.view(self.num_modes, -1, self.future_steps, 2), min=0.05)
        return torch.cat((loc, scale), dim=-1), pi
