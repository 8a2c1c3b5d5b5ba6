"""Global interactor: three rounds of agent-agent edge attention over one shared edge set
(``models/global_interactor.py``), on the same kernels as the local encoder.

Per layer the key / value of an edge is ``lin_*_node(x[src]) + lin_*_edge(rel_embed[e])``.  The node
terms are projected once per NODE ([N, 2D]) and only the edge terms are projected per edge (one fused
[E, D] x [D, 2D] product whose store epilogue adds the source node's row), which is exactly equal to
projecting the lifted rows (``global_interactor.py:115-119``).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .embedding import MultipleInputEmbedding, SingleInputEmbedding, init_weights
from .local_encoder import (_edge_set_from_coo, _feed_forward, _fused_linear, _linear, _pre_norm_ffn,
                            build_edge_plan)
from .ops import EdgeSet


class GlobalInteractorLayer(nn.Module):
    """``models/global_interactor.py:68-142``."""

    def __init__(self, embed_dim: int, num_heads: int = 8, dropout: float = 0.1, **kwargs) -> None:
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.lin_q_node = nn.Linear(embed_dim, embed_dim)
        self.lin_k_node = nn.Linear(embed_dim, embed_dim)
        self.lin_k_edge = nn.Linear(embed_dim, embed_dim)
        self.lin_v_node = nn.Linear(embed_dim, embed_dim)
        self.lin_v_edge = nn.Linear(embed_dim, embed_dim)
        self.lin_self = nn.Linear(embed_dim, embed_dim)
        self.attn_drop = nn.Dropout(dropout)
        self.lin_ih = nn.Linear(embed_dim, embed_dim)
        self.lin_hh = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.proj_drop = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.mlp = nn.Sequential(nn.Linear(embed_dim, embed_dim * 4), nn.ReLU(inplace=True), nn.Dropout(dropout),
                                 nn.Linear(embed_dim * 4, embed_dim), nn.Dropout(dropout))

    def paired_layers(self):
        """Layers evaluated as one product (``HiVT.pack_parameters`` lays their parameters back to back)."""
        return [(self.lin_q_node, self.lin_k_node, self.lin_v_node, self.lin_hh, self.lin_self),
                (self.lin_k_edge, self.lin_v_edge)]

    def forward(self, x: torch.Tensor, edge_index: Optional[torch.Tensor], edge_attr: torch.Tensor, size=None,
                edges: Optional[EdgeSet] = None, edge_grad_sink=None) -> torch.Tensor:
        """With ``edges`` given, ``edge_attr`` [E, D] is already in its CSR order; otherwise
        ``edge_index`` / ``edge_attr`` are the reference's COO list and per-edge embeddings.  ``edge_grad_sink``:
        ``edge_attr`` came from ``ops.share_edge_rows`` (several layers read it)."""
        if edges is None:
            edges = _edge_set_from_coo(edge_index, edge_attr.new_zeros(edge_attr.shape[0], 2), x.shape[0], x.shape[0])
            edges.src = edges.coo_src.index_select(0, edges.perm.long())
            edge_attr = edge_attr.index_select(0, edges.perm.long())
        normed, x = ops.layer_norm_residual(x, self.norm1)
        x = self._mha_block(normed, edges, edge_attr, residual=x, edge_grad_sink=edge_grad_sink)
        return _pre_norm_ffn(x, self.norm2, self.mlp)

    def _mha_block(self, x: torch.Tensor, edges: EdgeSet, edge_attr: torch.Tensor,
                   residual: Optional[torch.Tensor] = None, edge_grad_sink=None) -> torch.Tensor:
        edges.group_by_source()
        if ops.attention_block_supported(x, edge_attr, residual) and edges.src.dtype == torch.int32:
            return ops.attention_block(x, residual, edge_attr, edges.rowptr, self.lin_q_node, self.lin_k_edge,
                                       self.lin_v_edge, self.lin_ih, self.lin_hh, self.lin_self, self.out_proj,
                                       self.num_heads, self.attn_drop.p, self.proj_drop.p, self.training,
                                       tag=type(self).__name__, padded=edges.count is not None,
                                       lin_k_node=self.lin_k_node, lin_v_node=self.lin_v_node, src=edges.src,
                                       rowptr_by_src=edges.rowptr_by_src, perm_by_src=edges.perm_by_src,
                                       edge_grad_sink=edge_grad_sink)
        q = _linear(x, self.lin_q_node)
        kv_node = _fused_linear(x, self.lin_k_node, self.lin_v_node)
        if edge_attr.is_cuda and edge_attr.dtype == torch.float32 and edges.src.dtype == torch.int32:
            # key | value of an edge = edge terms + its source node's terms: the node rows are added where the edge
            # projection stores its tiles, and the attention kernels stream finished rows (no gathers in them)
            kv = ops.pair_linear(edge_attr, self.lin_k_edge, self.lin_v_edge, table=kv_node, index=edges.src,
                                 rowptr_by_index=edges.rowptr_by_src, perm_by_index=edges.perm_by_src)
            agg = ops.seg_attn(q, kv, edges.rowptr, self.num_heads, self.attn_drop.p, self.training,
                               tag=type(self).__name__, padded=edges.count is not None)
        else:
            kv_edge = _fused_linear(edge_attr, self.lin_k_edge, self.lin_v_edge)
            agg = ops.seg_attn(q, kv_edge, edges.rowptr, self.num_heads, self.attn_drop.p, self.training,
                               kv_node=kv_node, src=edges.src, rowptr_by_src=edges.rowptr_by_src,
                               perm_by_src=edges.perm_by_src, tag=type(self).__name__,
                               padded=edges.count is not None)
        if residual is not None:
            return ops.gated_update_tail(agg, x, residual, self.lin_ih, self.lin_hh, self.lin_self, self.out_proj,
                                         self.proj_drop.p, self.proj_drop.training)
        upd = ops.gated_update(agg, _linear(agg, self.lin_ih), _fused_linear(x, self.lin_hh, self.lin_self))
        return _linear(upd, self.out_proj, drop=self.proj_drop, residual=residual)

    def _ff_block(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        return _feed_forward(self.mlp, x, residual)


class GlobalInteractor(nn.Module):
    """``models/global_interactor.py:18-65``."""

    def __init__(self, historical_steps: int, embed_dim: int, edge_dim: int, num_modes: int = 6, num_heads: int = 8,
                 num_layers: int = 3, dropout: float = 0.1, rotate: bool = True) -> None:
        super().__init__()
        self.historical_steps = historical_steps
        self.embed_dim = embed_dim
        self.num_modes = num_modes
        if rotate:
            self.rel_embed = MultipleInputEmbedding(in_channels=[edge_dim, edge_dim], out_channel=embed_dim)
        else:
            self.rel_embed = SingleInputEmbedding(in_channel=edge_dim, out_channel=embed_dim)
        self.global_interactor_layers = nn.ModuleList(
            [GlobalInteractorLayer(embed_dim=embed_dim, num_heads=num_heads, dropout=dropout)
             for _ in range(num_layers)])
        self.norm = nn.LayerNorm(embed_dim)
        self.multihead_proj = nn.Linear(embed_dim, num_modes * embed_dim)
        self.apply(init_weights)

    def forward(self, data, local_embed: torch.Tensor) -> torch.Tensor:
        plan = data.get("edge_plan")
        if plan is None:
            plan = build_edge_plan(data, self.historical_steps, None)
            data["edge_plan"] = plan
        edges = plan.interaction
        if data["rotate_mat"] is None:
            # un-rotated relative positions in CSR order (global_interactor.py:51-52)
            feat, src_csr, _ = ops.edge_gather_rotate(edges.coo_attr.new_zeros(edges.num_src, 2), edges.coo_attr, None,
                                                      edges.coo_src, edges.coo_dst, edges.perm, 1, edges.count)
            rel_embed = self.rel_embed(feat[:, 2:4])
        else:
            feat, src_csr, _ = ops.edge_relpose(edges.coo_attr, data["rotate_mat"], data["rotate_angles"],
                                                edges.coo_src, edges.coo_dst, edges.perm, edges.count)
            rel_embed = self.rel_embed.forward_packed(feat)
        edges.src = src_csr
        x = local_embed
        sink = None
        if (len(self.global_interactor_layers) > 1 and rel_embed.requires_grad and torch.is_grad_enabled()
                and ops.attention_block_supported(x, rel_embed, x) and src_csr.dtype == torch.int32):
            # every layer reads the embeddings: their gradient is summed inside the layers' own products
            rel_embed, sink = ops.share_edge_rows(rel_embed)
        for layer in self.global_interactor_layers:
            x = layer(x, None, rel_embed, edges=edges, edge_grad_sink=sink)
        x = ops.layer_norm(x, self.norm)
        if x.is_cuda and x.dtype == torch.float32:
            x = ops.fused_linear(x, self.multihead_proj.weight, self.multihead_proj.bias)
        else:
            x = self.multihead_proj(x)
        x = x.view(-1, self.num_modes, self.embed_dim)
        return x.transpose(0, 1)
