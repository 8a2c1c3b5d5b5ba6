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

from . import ops


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
        e = self.embed  # Linear, LN, ReLU, Linear, LN, ReLU, Linear, LN -> three fused launches
        if x.is_cuda and ops.embed2_supported([e[0].in_features], e[0].out_features):
            h = ops.embed2(x, [(e[0].weight, e[0].bias, e[1].weight, e[1].bias)], e[1].eps)
        else:
            h = ops.fused_linear(x, e[0].weight, e[0].bias, layer_norm=e[1], relu=True)
        h = ops.fused_linear(h, e[3].weight, e[3].bias, layer_norm=e[4], relu=True)
        return ops.fused_linear(h, e[6].weight, e[6].bias, layer_norm=e[7], relu=False)


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

    def _fuse(self, branch_inputs: Sequence[torch.Tensor], extra: Optional[torch.Tensor],
              packed: Optional[torch.Tensor] = None, extra_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Branch first layers (Linear+LN+ReLU; one streaming launch for all branches when every input
        has two components, else one launch each), then ALL second linears as one product over the
        concatenated hidden rows ([h_0 | h_1 | ..] @ [W_0 | W_1 | ..]^T + sum of biases, which equals
        the sum of the branch outputs), the categorical rows added in the same epilogue as LN+ReLU,
        then Linear+LN.  ``packed`` holds the inputs side by side if the caller has them that way."""
        width = self.module_list[0][0].out_features
        if branch_inputs[0].is_cuda and ops.embed2_supported(self.in_channels, width):
            if packed is None:
                packed = branch_inputs[0] if len(branch_inputs) == 1 else torch.cat(list(branch_inputs), dim=-1)
            h = ops.embed2(packed, [(br[0].weight, br[0].bias, br[1].weight, br[1].bias) for br in self.module_list],
                           self.module_list[0][1].eps)
        else:
            hidden = [ops.fused_linear(inp, br[0].weight, br[0].bias, layer_norm=br[1], relu=True)
                      for br, inp in zip(self.module_list, branch_inputs)]
            h = hidden[0] if len(hidden) == 1 else torch.cat(hidden, dim=-1)
        a = self.aggr_embed
        if len(self.module_list) == 2 and h.is_cuda:
            # (the two layers' own parameters go down: their gradients come back per layer, as leaves of the graph)
            first, second = self.module_list[0][3], self.module_list[1][3]
            # (the weights side by side, if the model has laid them out for this pass already: used once)
            stacked = self.__dict__.pop("_stacked_pair", None)
            h = ops.fused_linear(h, first.weight, first.bias, add=extra, layer_norm=a[0], relu=True, add_index=extra_index,
                                 weight_b=second.weight, bias_b=second.bias, stacked=stacked)
            return ops.fused_linear(h, a[2].weight, a[2].bias, layer_norm=a[3], relu=False)
        weight = self.module_list[0][3].weight if len(self.module_list) == 1 else \
            torch.cat([br[3].weight for br in self.module_list], dim=1)
        bias = self.module_list[0][3].bias
        for br in self.module_list[1:]:
            bias = bias + br[3].bias
        h = ops.fused_linear(h, weight, bias, add=extra, layer_norm=a[0], relu=True, add_index=extra_index)
        return ops.fused_linear(h, a[2].weight, a[2].bias, layer_norm=a[3], relu=False)

    def summed_pair(self):
        """``(w_a, b_a, w_b, b_b)`` of the two branches' second Linears if the module has two branches whose outputs
        are summed as one product, else None."""
        if len(self.module_list) != 2:
            return None
        first, second = self.module_list[0][3], self.module_list[1][3]
        return first.weight, first.bias, second.weight, second.bias

    def forward(self, continuous_inputs: List[torch.Tensor],
                categorical_inputs: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        cat = None
        if categorical_inputs is not None:
            cat = categorical_inputs[0]
            for extra in categorical_inputs[1:]:
                cat = cat + extra
        return self._fuse(continuous_inputs, cat)

    def forward_packed(self, packed: torch.Tensor, categorical_sum: Optional[torch.Tensor] = None,
                       categorical_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Same computation for inputs packed side by side in one [E, sum(in_channels)] record
        (the layout the gather kernel emits).  With ``categorical_index`` (int32 per row),
        ``categorical_sum`` is a small table of already summed categorical embeddings and row e adds
        ``categorical_sum[categorical_index[e]]``."""
        parts, col = [], 0
        for width in self.in_channels:
            parts.append(packed[:, col:col + width])
            col += width
        return self._fuse(parts, categorical_sum, packed if col == packed.shape[-1] else None, categorical_index)
