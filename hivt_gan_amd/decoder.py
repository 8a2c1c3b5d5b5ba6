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
    mods = list(seq)
    if not x.is_cuda:
        for mod in mods:
            x = mod(x)
        return x
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

    def _forward_fused(self, local_embed: torch.Tensor, global_embed: torch.Tensor
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
        """The GPU route: no concatenated inputs (``ops.decoder_inputs``), the rows kept in the order they arrive in,
        and the output tensor assembled by one launch (``ops.decoder_output``)."""
        modes, n, d = global_embed.shape
        # [F, N, D] as a transposed view of [N, F, D] (what the interaction module hands over): rows stay
        # agent-major, which saves the copy into mode-major order and leaves pi [N, F] contiguous
        agent_major = modes > 1 and global_embed.transpose(0, 1).is_contiguous()
        g = global_embed.transpose(0, 1).reshape(n * modes, d) if agent_major else global_embed.reshape(modes * n, d)
        out, pi_h = ops.decoder_inputs(g, local_embed, modes, agent_major, self.aggr_embed[0], self.aggr_embed[1],
                                       self.pi[0], self.pi[1])
        pi = _run(list(self.pi)[3:], pi_h)
        pi = pi.view(n, modes) if agent_major else pi.view(modes, n).t()
        loc = _run(self.loc, out)
        if not self.uncertain:
            loc = loc.view(n, modes, self.future_steps, 2).transpose(0, 1) if agent_major else \
                loc.view(modes, n, self.future_steps, 2)
            return loc, pi
        return ops.decoder_output(loc, _run(self.scale, out), modes, agent_major, 0.05), pi

    def forward(self, local_embed: torch.Tensor, global_embed: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """local_embed [N, D], global_embed [F, N, D] -> (y_hat [F, N, H, 4 or 2], pi [N, F])."""
        if (local_embed.is_cuda and local_embed.dtype == torch.float32 and self.input_size == self.hidden_size
                and self.hidden_size <= 128 and self.hidden_size % 4 == 0 and global_embed.shape[0] == self.num_modes
                and local_embed.numel() > 0):
            return self._forward_fused(local_embed, global_embed)
        tiled = local_embed.expand(self.num_modes, *local_embed.shape)
        pi = _run(self.pi, torch.cat((tiled, global_embed), dim=-1)).squeeze(-1).t()
        out = _run(self.aggr_embed, torch.cat((global_embed, tiled), dim=-1))
        loc = _run(self.loc, out).view(self.num_modes, -1, self.future_steps, 2)
        if not self.uncertain:
            return loc, pi
        scale = F.elu(_run(self.scale, out), alpha=1.0) + 1.0
        scale = torch.clamp(scale.view(self.num_modes, -1, self.future_steps, 2), min=0.05)
        return torch.cat((loc, scale), dim=-1), pi
