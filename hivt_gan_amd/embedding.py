"""Embedding MLPs and weight initialisation with the reference's parameter names.

``SingleInputEmbedding`` / ``MultipleInputEmbedding`` keep the state_dict layout of
``models/embedding.py:9-57`` (``embed.{0,1,3,4,6,7}``, ``module_list.{i}.{0,1,3}``,
``aggr_embed.{0,2,3}``); ``init_weights`` applies the table of ``utils.py:81-146`` for the layer
types that occur in the model.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn


def init_weights(m: nn.Module) -> None:
    """Xavier-uniform linears with zero bias, unit LayerNorm, N(0, 0.02) embeddings,
    Xavier-style attention projections (``utils.py:81-119``)."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Embedding):
        nn.init.normal_(m.weight, mean=0.0, std=0.02)
    elif isinstance(m, nn.MultiheadAttention):
        if m.in_proj_weight is not None:
            bound = (6.0 / (m.embed_dim + m.embed_dim)) ** 0.5
            nn.init.uniform_(m.in_proj_weight, -bound, bound)
        else:
            for w in (m.q_proj_weight, m.k_proj_weight, m.v_proj_weight):
                nn.init.xavier_uniform_(w)
        if m.in_proj_bias is not None:
            nn.init.zeros_(m.in_proj_bias)
        nn.init.xavier_uniform_(m.out_proj.weight)
        if m.out_proj.bias is not None:
            nn.init.zeros_(m.out_proj.bias)
        for b in (m.bias_k, m.bias_v):
            if b is not None:
                nn.init.normal_(b, mean=0.0, std=0.02)


def _branch(in_channel: int, out_channel: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(in_channel, out_channel), nn.LayerNorm(out_channel), nn.ReLU(inplace=True),
                         nn.Linear(out_channel, out_channel))


class SingleInputEmbedding(nn.Module):
    """Three Linear+LayerNorm stages with ReLU between them (``models/embedding.py:9-27``)."""

    def __init__(self, in_channel: int, out_channel: int) -> None:
        super().__init__()
        stages: List[nn.Module] = []
        width = in_channel
        for i in range(3):
            stages += [nn.Linear(width, out_channel), nn.LayerNorm(out_channel)]
            if i < 2:
                stages.append(nn.ReLU(inplace=True))
            width = out_channel
        self.embed = nn.Sequential(*stages)
        self.apply(init_weights)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.embed(x)


class MultipleInputEmbedding(nn.Module):
    """Per-input branch MLPs, summed (plus categorical rows), then LN-ReLU-Linear-LN
    (``models/embedding.py:30-57``)."""

    def __init__(self, in_channels: Sequence[int], out_channel: int) -> None:
        super().__init__()
        self.in_channels = list(in_channels)
        self.module_list = nn.ModuleList([_branch(c, out_channel) for c in in_channels])
        self.aggr_embed = nn.Sequential(nn.LayerNorm(out_channel), nn.ReLU(inplace=True),
                                        nn.Linear(out_channel, out_channel), nn.LayerNorm(out_channel))
        self.apply(init_weights)

    def forward(self, continuous_inputs: List[torch.Tensor],
                categorical_inputs: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        total = None
        for branch, inp in zip(self.module_list, continuous_inputs):
            h = branch(inp)
            total = h if total is None else total + h
        if categorical_inputs is not None:
            cat = categorical_inputs[0]
            for extra in categorical_inputs[1:]:
                cat = cat + extra
            total = total + cat
        return self.aggr_embed(total)

    def forward_packed(self, packed: torch.Tensor, categorical_sum: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Same computation for inputs packed side by side in one [E, sum(in_channels)] record
        (the layout the gather kernel emits)."""
        total, col = None, 0
        for branch, width in zip(self.module_list, self.in_channels):
            h = branch(packed[:, col:col + width])
            total = h if total is None else total + h
            col += width
        if categorical_sum is not None:
            total = total + categorical_sum
        return self.aggr_embed(total)
