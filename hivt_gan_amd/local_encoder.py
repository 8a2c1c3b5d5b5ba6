"""Local encoder: agent-agent (AAEncoder) and agent-lane (ALEncoder) edge attention, the temporal
transformer between them, and the wrapper that builds every edge set of a batch on the device.

The modules keep the constructors, forward signatures and state_dict keys of
``models/local_encoder.py`` so callers and checkpoints are interchangeable, but none of them uses
torch_geometric: message / aggregate / update run as

    edge gather+rotate kernel  ->  per-edge embedding MLP (dense)  ->  fused K|V projection
    ->  segment-softmax attention kernel over CSR rows  ->  gated-update kernel

with Q hoisted from per-edge to per-node (``lin_q(c)[dst] == lin_q(c[dst])``).  The 20 timesteps of
the agent-agent stage always run as ONE batched call (node id ``t*N + n``), which is the arithmetic
of the reference's ``parallel=True`` branch (``local_encoder.py:62-71,135-144,168-169``) and, row
for row, of its per-timestep loop (``:73-78``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .embedding import MultipleInputEmbedding, SingleInputEmbedding, init_weights
from .ops import EdgeSet


# ----------------------------------------------------------------------------------------------
# Edge plan: every edge set of one batch, built with one host read-back of the three edge counts
# ----------------------------------------------------------------------------------------------
@dataclass
class EdgePlan:
    agent_agent: EdgeSet   # all timesteps, node id t*N + n        (local_encoder.py:58-61,65)
    lane_actor: EdgeSet    # lane -> actor within the radius        (local_encoder.py:80)
    interaction: EdgeSet   # agents valid at the last history step  (global_interactor.py:48-50)
    overflow: Optional[torch.Tensor] = None  # device bool: a filtered list exceeded its fixed capacity
    counts: Optional[tuple] = None           # three device int32 scalars: unclamped live counts (fixed-capacity mode)
    # the (step, agent) nodes of the agent-agent stage that are not padded, packed: the stage's output at a padded
    # node is replaced by the padding token before anything reads it (local_encoder.py:230), so the stage is
    # evaluated for the other nodes only
    aa_rows: Optional[ops.RowSubset] = None


class DistanceDropEdge:
    """Radius filter with the call signature of ``utils.py:64-78`` (device-side compaction)."""

    def __init__(self, max_distance: Optional[float] = None) -> None:
        self.max_distance = max_distance

    def __call__(self, edge_index: torch.Tensor, edge_attr: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.max_distance is None:
            return edge_index, edge_attr
        src, dst, attr, count = ops.edges_radius(edge_index, edge_attr, self.max_distance)
        e = int(count.item())
        return torch.stack([src[:e], dst[:e]]).long(), attr[:e]


def subgraph(subset: torch.Tensor, edge_index: torch.Tensor) -> Tuple[torch.Tensor, None]:
    """Edges whose two endpoints are in the boolean ``subset`` (``torch_geometric.utils.subgraph`` as
    called at ``local_encoder.py:59`` / ``global_interactor.py:48``), order preserving."""
    n = subset.shape[0]
    pm = (~subset).view(n, 1)
    pos = torch.zeros(n, 1, 2, dtype=torch.float32, device=subset.device)
    src, dst, _attr, count = ops.edges_temporal(edge_index, pm, pos, 0, 1, None)
    e = int(count.item())
    return torch.stack([src[:e], dst[:e]]).long(), None


# The agent-agent stage is evaluated for the unpadded (step, agent) nodes only; False evaluates every node, as
# the reference does (the results that anything reads are the same either way).
PACK_AGENT_ROWS = True


def _pair_sets(data, T: int, n: int, local_radius: Optional[float], caps):
    """Agent-agent sets (all history steps, last-step interaction) straight in CSR order from the agents
    themselves; ``caps`` = capacities of the two lists."""
    out = []
    for (t0, steps, radius), cap in zip(((0, T, local_radius), (T - 1, 1, None)), caps):
        rowptr, src, dst, attr, count = ops.edges_all_pairs_csr(data["scene_range"], data["padding_mask"],
                                                                data["positions"], t0, steps, radius, cap)
        out.append((rowptr, src, dst, attr, count, steps * n))
    return out


def build_edge_plan(data, historical_steps: int, local_radius: Optional[float], from_pairs: Optional[bool] = None
                    ) -> EdgePlan:
    """All three edge sets of a batch, filtered and CSR-grouped on the device.

    Without ``data["edge_capacity"]`` the three live counts are read back once and the lists are cut
    to size.  With it (three integers, see ``BatchSpec``) nothing is read back: the lists keep their
    fixed capacity, every consumer takes the live count from device memory, and ``overflow`` records on
    the device whether a filtered list did not fit (the step must then be discarded).

    Batches whose candidate list is every ordered pair of a scene's agents (``data["all_pairs"]``, set by
    ``collate``) get their two agent-agent sets from ``hivt_edges_all_pairs_csr``: no candidate list is
    read and nothing is sorted; other batches go through the order-preserving filter of the given list
    and a stable sort by target.  Both give the same sets in the same order.  ``from_pairs`` forces one
    route (tests)."""
    T = historical_steps
    n = int(data.num_nodes)
    num_lanes = int(data["lane_vectors"].shape[0])
    caps = data.get("edge_capacity")
    if from_pairs is None:
        from_pairs = bool(data.get("all_pairs", False)) and "scene_range" in data
    # nodes (t, n) of the agent-agent stage that are not padded, t-major (the fixed-capacity all-pairs route reads
    # the mask itself and only needs to know that the rows are packed)
    pack = PACK_AGENT_ROWS and n > 0
    fused_rows = pack and caps is not None and from_pairs and data["padding_mask"].is_cuda
    valid = None
    if pack:
        valid = True if fused_rows else ~data["padding_mask"][:, :T].t().reshape(-1)
    # (read back with the edge counts when no capacities are given; a fixed-capacity plan never needs it)
    live_rows = valid.sum(dtype=torch.int32).view(1) if (valid is not None and caps is None) else None
    # (of the three sets only the lane -> actor rows differ widely in length -- an agent near an intersection has
    # several times the lane segments of one on a straight road -- so only that set asks for the ranking of its rows)
    la = ops.edges_radius(data["lane_actor_index"], data["lane_actor_vectors"], local_radius)
    row_cap = min(int(caps[3]), T * n) if (caps is not None and len(caps) > 3) else T * n
    if from_pairs:
        e0 = int(data["edge_index"].shape[1])
        pair_caps = (min(int(caps[0]), e0 * T), min(int(caps[2]), e0)) if caps is not None else (e0 * T, e0)
        aa, gi = _pair_sets(data, T, n, local_radius, pair_caps)
        if caps is None:
            # the only host synchronisation of the forward pass
            counts = torch.cat([aa[4], la[3], gi[4]] + ([live_rows] if valid is not None else [])).cpu()
            c_aa, c_la, c_gi = (int(c) for c in counts[:3])
            sets = [EdgeSet(T * n, T * n, aa[1][:c_aa], aa[2][:c_aa], aa[3][:c_aa], aa[0], None, aa[1][:c_aa], aa[2][:c_aa]),
                    ops.make_edge_set(la[0][:c_la], la[1][:c_la], la[2][:c_la], num_lanes, n, order=True),
                    EdgeSet(n, n, gi[1][:c_gi], gi[2][:c_gi], gi[3][:c_gi], gi[0], None, gi[1][:c_gi], gi[2][:c_gi])]
            rows = ops.row_subset(valid, None, sets[0].rowptr, int(counts[3])) if valid is not None else None
            return EdgePlan(*sets, aa_rows=rows)
        c_la = min(int(caps[1]), int(la[0].shape[0]))
        if valid is not None:
            # every consumer of a fixed-capacity list bounds the device count by the list's length itself, so the
            # counts go down as they are; the packed agent-agent rows and the overflow flag of all four capacities
            # come out of one call
            sets = [EdgeSet(T * n, T * n, aa[1], aa[2], aa[3], aa[0], None, aa[1], aa[2], count=aa[4]),
                    ops.make_edge_set(la[0][:c_la], la[1][:c_la], la[2][:c_la], num_lanes, n, count=la[3], order=True),
                    EdgeSet(n, n, gi[1], gi[2], gi[3], gi[0], None, gi[1], gi[2], count=gi[4])]
            rows = ops.row_subset_fixed(data["padding_mask"], T, row_cap, sets[0].rowptr, (aa[4], la[3], gi[4]),
                                        (pair_caps[0], c_la, pair_caps[1]))
            return EdgePlan(*sets, overflow=rows.overflow, counts=(aa[4], la[3], gi[4]), aa_rows=rows)
        sets = [EdgeSet(T * n, T * n, aa[1], aa[2], aa[3], aa[0], None, aa[1], aa[2], count=aa[4].clamp(max=pair_caps[0])),
                ops.make_edge_set(la[0][:c_la], la[1][:c_la], la[2][:c_la], num_lanes, n, count=la[3].clamp(max=c_la),
                                  order=True),
                EdgeSet(n, n, gi[1], gi[2], gi[3], gi[0], None, gi[1], gi[2], count=gi[4].clamp(max=pair_caps[1]))]
        overflow = (aa[4] > pair_caps[0]) | (la[3] > c_la) | (gi[4] > pair_caps[1])
        return EdgePlan(*sets, overflow=overflow.view(()), counts=(aa[4], la[3], gi[4]), aa_rows=None)
    aa = ops.edges_temporal(data["edge_index"], data["padding_mask"], data["positions"], 0, T, local_radius)
    gi = ops.edges_temporal(data["edge_index"], data["padding_mask"], data["positions"], T - 1, 1, None)
    shapes = ((T * n, T * n), (num_lanes, n), (n, n))
    if caps is None:
        # the only host synchronisation of the forward pass
        counts = torch.cat([aa[3], la[3], gi[3]] + ([live_rows] if valid is not None else [])).cpu()
        sets = [ops.make_edge_set(lst[0][:c], lst[1][:c], lst[2][:c], ns, nd, order=lst is la)
                for lst, c, (ns, nd) in zip((aa, la, gi), (int(c) for c in counts[:3]), shapes)]
        rows = ops.row_subset(valid, None, sets[0].rowptr, int(counts[3])) if valid is not None else None
        return EdgePlan(*sets, aa_rows=rows)
    caps = [min(int(c), int(lst[0].shape[0])) for c, lst in zip(caps, (aa, la, gi))]
    sets, overflow = [], None
    for lst, c, (ns, nd) in zip((aa, la, gi), caps, shapes):
        sets.append(ops.make_edge_set(lst[0][:c], lst[1][:c], lst[2][:c], ns, nd, count=lst[3].clamp(max=c),
                                      order=lst is la))
        over = lst[3] > c
        overflow = over if overflow is None else overflow | over
    rows = ops.row_subset(valid, row_cap, sets[0].rowptr) if valid is not None else None
    if rows is not None:
        overflow = overflow | rows.overflow
    return EdgePlan(*sets, overflow=overflow.view(()), counts=(aa[3], la[3], gi[3]), aa_rows=rows)


def _edge_set_from_coo(edge_index: torch.Tensor, edge_attr: torch.Tensor, num_src: int, num_dst: int) -> EdgeSet:
    src = edge_index[0].to(torch.int32).contiguous()
    dst = edge_index[1].to(torch.int32).contiguous()
    return ops.make_edge_set(src, dst, edge_attr.contiguous(), num_src, num_dst)


def _fused_linear(x: torch.Tensor, a: nn.Linear, b: nn.Linear) -> torch.Tensor:
    """[a(x) | b(x)] as one product."""
    if x.is_cuda and x.dtype == torch.float32:
        return ops.pair_linear(x, a, b)  # no copy of the parameters once they are packed (HiVT.pack_parameters)
    return ops.fused_linear(x, torch.cat([a.weight, b.weight], dim=0), torch.cat([a.bias, b.bias], dim=0))


def _linear(x: torch.Tensor, lin: nn.Linear, relu: bool = False, drop: Optional[nn.Dropout] = None,
            residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[residual +] [drop]([relu](lin(x)))`` in one launch."""
    if drop is None:
        return ops.fused_linear(x, lin.weight, lin.bias, relu=relu, residual=residual)
    return ops.fused_linear(x, lin.weight, lin.bias, relu=relu, dropout_p=drop.p, training=drop.training,
                            residual=residual)


def _feed_forward(mlp: nn.Sequential, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[residual +]`` Linear, ReLU, Dropout, Linear, Dropout (``local_encoder.py:117-122``): two launches."""
    return _linear(_linear(x, mlp[0], relu=True, drop=mlp[2]), mlp[3], drop=mlp[4], residual=residual)


def _pre_norm_ffn(x: torch.Tensor, norm: nn.LayerNorm, mlp: nn.Sequential) -> torch.Tensor:
    """``x + mlp(norm(x))`` for the Linear, ReLU, Dropout, Linear, Dropout stacks: one autograd node on
    the GPU, framework ops elsewhere."""
    if ops.pre_norm_feed_forward_supported(x, mlp[0], mlp[3]):
        return ops.pre_norm_feed_forward(x, norm, mlp[0], mlp[2].p, mlp[3], mlp[4].p, mlp[2].training)
    normed, x = ops.layer_norm_residual(x, norm)
    return _feed_forward(mlp, normed, x)


class _EdgeAttentionBase(nn.Module):
    """Parameters and the dense tail shared by AAEncoder and ALEncoder."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float) -> None:
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.dropout = dropout
        self.lin_q = nn.Linear(embed_dim, embed_dim)
        self.lin_k = nn.Linear(embed_dim, embed_dim)
        self.lin_v = nn.Linear(embed_dim, embed_dim)
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
        return [(self.lin_q, self.lin_hh, self.lin_self), (self.lin_k, self.lin_v)]

    def _attend_and_update(self, center: torch.Tensor, edge_embed: torch.Tensor, edges: EdgeSet,
                           residual: Optional[torch.Tensor] = None, rowptr: Optional[torch.Tensor] = None
                           ) -> torch.Tensor:
        """``[residual +]`` message (Q.K, segment softmax, weighting) + aggregate + update + out_proj +
        dropout for per-edge embeddings already in CSR order (``local_encoder.py:174-188,197-199`` /
        ``:364-379,391-394``)."""
        # `rowptr`: the edge set's row pointer re-indexed to packed target rows (`center` then holds those rows)
        # (a set with widely spread row lengths brings the order its attention launches start their rows in)
        order = edges.row_order if rowptr is None else None
        rowptr = edges.rowptr if rowptr is None else rowptr
        if ops.attention_block_supported(center, edge_embed, residual):
            return ops.attention_block(center, residual, edge_embed, rowptr, self.lin_q, self.lin_k, self.lin_v,
                                       self.lin_ih, self.lin_hh, self.lin_self, self.out_proj, self.num_heads,
                                       self.attn_drop.p, self.proj_drop.p, self.training, tag=type(self).__name__,
                                       padded=edges.count is not None, row_order=order)
        q = _linear(center, self.lin_q)
        if edge_embed.is_cuda and edge_embed.dtype == torch.float32 and edge_embed.dim() == 2:
            # key | value projection and attention as one node: their shared gradient stays inside it
            agg = ops.edge_attention(q, edge_embed, self.lin_k, self.lin_v, rowptr, self.num_heads, self.attn_drop.p,
                                     self.training, tag=type(self).__name__, padded=edges.count is not None,
                                     row_order=order)
        else:
            kv = _fused_linear(edge_embed, self.lin_k, self.lin_v)
            agg = ops.seg_attn(q, kv, rowptr, self.num_heads, self.attn_drop.p, self.training,
                               tag=type(self).__name__, padded=edges.count is not None, row_order=order)
        if residual is not None:
            return ops.gated_update_tail(agg, center, residual, self.lin_ih, self.lin_hh, self.lin_self, self.out_proj,
                                         self.proj_drop.p, self.proj_drop.training)
        upd = ops.gated_update(agg, _linear(agg, self.lin_ih), _fused_linear(center, self.lin_hh, self.lin_self))
        return _linear(upd, self.out_proj, drop=self.proj_drop, residual=residual)

    def _ff_block(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        return _feed_forward(self.mlp, x, residual)


class AAEncoder(_EdgeAttentionBase):
    """Agent-agent edge attention of one timestep or of all timesteps batched
    (``models/local_encoder.py:87-202``)."""

    def __init__(self, historical_steps: int, node_dim: int, edge_dim: int, embed_dim: int, num_heads: int = 8,
                 dropout: float = 0.1, parallel: bool = False, **kwargs) -> None:
        super().__init__(embed_dim, num_heads, dropout)
        self.historical_steps = historical_steps
        self.parallel = parallel
        self.center_embed = SingleInputEmbedding(in_channel=node_dim, out_channel=embed_dim)
        self.nbr_embed = MultipleInputEmbedding(in_channels=[node_dim, edge_dim], out_channel=embed_dim)
        self.bos_token = nn.Parameter(torch.empty(historical_steps, embed_dim))
        nn.init.normal_(self.bos_token, mean=0.0, std=0.02)
        self.apply(init_weights)

    def forward(self, x: torch.Tensor, t: Optional[int], edge_index: Optional[torch.Tensor],
                edge_attr: Optional[torch.Tensor], bos_mask: torch.Tensor, rotate_mat: Optional[torch.Tensor] = None,
                size=None, edges: Optional[EdgeSet] = None, rows: Optional[ops.RowSubset] = None,
                x_rotated: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``parallel`` modules (or ``t is None``) take x [T*N, 2], bos_mask [N, T] and node ids
        ``t*N + n``; otherwise x [N, 2], an integer ``t`` and bos_mask [N].  ``edges`` may carry the
        already grouped edge set instead of ``edge_index`` / ``edge_attr``.  With ``rows`` (batched form, GPU) the
        stage is evaluated for that subset of the T*N nodes only and the result holds its packed rows
        [rows.capacity, D]; nodes outside the subset must own no edges.  ``x_rotated`` (with ``rows``): ``x`` already
        rotated into the agents' frames, [T*N, 2]."""
        T = self.historical_steps
        batched = self.parallel or t is None
        rowptr = None
        if rows is not None and not (batched and edges is not None and rows.rowptr is not None):
            raise ValueError("a row subset goes with the batched form and a grouped edge set")
        if batched and rows is not None and x_rotated is not None and self.embed_dim % 4 == 0:
            # packed nodes from prepared rows: one launch for the inputs of the centre embedding and the
            # begin-of-sequence codes, one for the substitution of the tokens
            n = x.shape[0] // T
            xc, code = ops.center_rows(x_rotated, bos_mask, rows.rows, n, T)
            center = ops.rows_put_tokens(self.center_embed(xc), code, self.bos_token)
            rowptr = rows.rowptr
            rot_mod = n
        elif batched:
            n = x.shape[0] // T
            xv = x.view(T, n, -1)
            if rotate_mat is not None:  # row vector times the agent's 2x2, written out (no batched GEMM)
                xv = xv[..., 0:1] * rotate_mat[:, 0, :] + xv[..., 1:2] * rotate_mat[:, 1, :]
            if rows is not None:
                step = torch.div(rows.rows, n, rounding_mode="floor")
                center = self.center_embed(xv.reshape(T * n, -1).index_select(0, rows.rows))
                center = ops.masked_row_tokens(center, bos_mask.t().reshape(-1).index_select(0, rows.rows), step,
                                               self.bos_token)
                rowptr = rows.rowptr
            else:
                center = self.center_embed(xv)
                if ops.step_tokens_supported(center, T):
                    center = ops.masked_step_tokens(center, bos_mask.t(), self.bos_token).reshape(x.shape[0], -1)
                else:
                    center = torch.where(bos_mask.t().unsqueeze(-1), self.bos_token.unsqueeze(-2),
                                         center).reshape(x.shape[0], -1)
            rot_mod = n
        else:
            xr = x if rotate_mat is None else x[:, 0:1] * rotate_mat[:, 0, :] + x[:, 1:2] * rotate_mat[:, 1, :]
            center = self.center_embed(xr)
            center = torch.where(bos_mask.unsqueeze(-1), self.bos_token[t], center)
            rot_mod = x.shape[0]
        if edges is None:
            edges = _edge_set_from_coo(edge_index, edge_attr, x.shape[0], x.shape[0])
        # the residual additions are part of the last launch of each block
        normed, center = ops.layer_norm_residual(center, self.norm1)
        center = self._mha_block(normed, x, edges, rotate_mat, rot_mod, residual=center, rowptr=rowptr)
        return _pre_norm_ffn(center, self.norm2, self.mlp)

    def _mha_block(self, center: torch.Tensor, x: torch.Tensor, edges: EdgeSet, rotate_mat: Optional[torch.Tensor],
                   rot_mod: int, residual: Optional[torch.Tensor] = None, rowptr: Optional[torch.Tensor] = None
                   ) -> torch.Tensor:
        # the per-edge features are looked up by the edges' own node ids in the full node set `x`, whether or not
        # `center` holds packed rows
        feat, _src, _dst = ops.edge_gather_rotate(x, edges.coo_attr, rotate_mat, edges.coo_src, edges.coo_dst,
                                                  edges.perm, rot_mod, edges.count)
        return self._attend_and_update(center, self.nbr_embed.forward_packed(feat), edges, residual, rowptr)


class ALEncoder(_EdgeAttentionBase):
    """Lane -> actor edge attention (``models/local_encoder.py:283-397``)."""

    def __init__(self, node_dim: int, edge_dim: int, embed_dim: int, num_heads: int = 8, dropout: float = 0.1,
                 **kwargs) -> None:
        super().__init__(embed_dim, num_heads, dropout)
        self.lane_embed = MultipleInputEmbedding(in_channels=[node_dim, edge_dim], out_channel=embed_dim)
        self.is_intersection_embed = nn.Parameter(torch.empty(2, embed_dim))
        self.turn_direction_embed = nn.Parameter(torch.empty(3, embed_dim))
        self.traffic_control_embed = nn.Parameter(torch.empty(2, embed_dim))
        for token in (self.is_intersection_embed, self.turn_direction_embed, self.traffic_control_embed):
            nn.init.normal_(token, mean=0.0, std=0.02)
        self.apply(init_weights)

    def forward(self, x: Tuple[torch.Tensor, torch.Tensor], edge_index: Optional[torch.Tensor],
                edge_attr: Optional[torch.Tensor], is_intersections: torch.Tensor, turn_directions: torch.Tensor,
                traffic_controls: torch.Tensor, rotate_mat: Optional[torch.Tensor] = None, size=None,
                edges: Optional[EdgeSet] = None, lane_code: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``lane_code`` (optional, int32 [L]): ``(is_intersections * 3 + turn_directions) * 2 + traffic_controls``
        if the caller has it already."""
        x_lane, x_actor = x
        if edges is None:
            edges = _edge_set_from_coo(edge_index, edge_attr, x_lane.shape[0], x_actor.shape[0])
        # one code per lane for the three categorical attributes, one table row per code
        code = lane_code if lane_code is not None else \
            (is_intersections.long() * 3 + turn_directions.long()) * 2 + traffic_controls.long()
        normed, x_actor = ops.layer_norm_residual(x_actor, self.norm1)
        x_actor = self._mha_block(normed, x_lane, edges, code, rotate_mat, residual=x_actor)
        return _pre_norm_ffn(x_actor, self.norm2, self.mlp)

    def _mha_block(self, x_actor: torch.Tensor, x_lane: torch.Tensor, edges: EdgeSet, lane_code: torch.Tensor,
                   rotate_mat: Optional[torch.Tensor], residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        feat, src_csr, _dst = ops.edge_gather_rotate(x_lane, edges.coo_attr, rotate_mat, edges.coo_src, edges.coo_dst,
                                                     edges.perm, x_actor.shape[0], edges.count)
        if x_actor.is_cuda and x_actor.dtype == torch.float32:
            table = ops.category_table(self.is_intersection_embed, self.turn_direction_embed, self.traffic_control_embed)
        else:
            table = ((self.is_intersection_embed.view(2, 1, 1, -1) + self.turn_direction_embed.view(1, 3, 1, -1))
                     + self.traffic_control_embed.view(1, 1, 2, -1)).reshape(12, -1)
        # the 12-row table is added per edge inside the embedding's second launch (row picked by the
        # lane's code); its gradient is a per-code sum of rows, not E*D contended atomic adds
        code_per_edge = lane_code.to(torch.int32).index_select(0, src_csr)
        return self._attend_and_update(x_actor, self.lane_embed.forward_packed(feat, table, code_per_edge), edges,
                                       residual)


# ----------------------------------------------------------------------------------------------
# Temporal encoder (dense; stock attention kernels)
# ----------------------------------------------------------------------------------------------
class TemporalEncoderLayer(nn.Module):
    """Pre-norm self-attention + feed-forward block (``models/local_encoder.py:244-280``)."""

    def __init__(self, embed_dim: int, num_heads: int = 8, dropout: float = 0.1) -> None:
        super().__init__()
        self.self_attn = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads, dropout=dropout)
        self.linear1 = nn.Linear(embed_dim, embed_dim * 4)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(embed_dim * 4, embed_dim)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)

    def _sa_block(self, x: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
        # residual + dropout1(out_proj(causal self-attention over the time axis)); x is [S, N, D]
        mha = self.self_attn
        S, N, D = x.shape
        H = mha.num_heads
        if x.is_cuda and ops.temporal_attention_supported(S, D, H):
            qkv = ops.fused_linear(x, mha.in_proj_weight, mha.in_proj_bias)
            o = ops.temporal_attention(qkv, H, mha.dropout, self.training)
            return ops.fused_linear(o, mha.out_proj.weight, mha.out_proj.bias, dropout_p=self.dropout1.p,
                                    training=self.training, residual=residual)
        if x.is_cuda:
            raise ops._unsupported(f"causal attention over {S} steps at width {D} with {H} heads "
                                   "(at most 32 steps; head widths 4, 8, 16, 32)")
        # CPU tensors: the framework's attention
        qkv = F.linear(x, mha.in_proj_weight, mha.in_proj_bias).view(S, N, 3, H, D // H)
        q, k, v = (qkv[:, :, i].permute(1, 2, 0, 3) for i in range(3))  # [N, H, S, d]
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=mha.dropout if self.training else 0.0, is_causal=True)
        o = o.permute(2, 0, 1, 3).reshape(S, N, D)
        return residual + self.dropout1(mha.out_proj(o))

    def _ff(self, x: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
        # residual + dropout2(linear2(dropout(relu(linear1(x)))))
        if x.is_cuda:
            h = ops.fused_linear(x, self.linear1.weight, self.linear1.bias, relu=True, dropout_p=self.dropout.p,
                                 training=self.training)
            return ops.fused_linear(h, self.linear2.weight, self.linear2.bias, dropout_p=self.dropout2.p,
                                    training=self.training, residual=residual)
        return residual + self.dropout2(self.linear2(self.dropout(F.relu(self.linear1(x)))))

    def _ff_residual(self, x: torch.Tensor) -> torch.Tensor:
        # x + _ff_block(norm2(x)) (``models/local_encoder.py:267``)
        if ops.pre_norm_feed_forward_supported(x, self.linear1, self.linear2):
            return ops.pre_norm_feed_forward(x, self.norm2, self.linear1, self.dropout.p, self.linear2, self.dropout2.p,
                                             self.training)
        normed, x = ops.layer_norm_residual(x, self.norm2)
        return self._ff(normed, x)

    def forward(self, src: torch.Tensor, src_mask: Optional[torch.Tensor] = None,
                src_key_padding_mask: Optional[torch.Tensor] = None, is_causal: bool = True) -> torch.Tensor:
        S, _, D = src.shape
        if src.is_cuda and src.dtype == torch.float32 and D <= 128 and src.numel() > 0 \
                and ops.temporal_attention_supported(S, D, self.self_attn.num_heads):
            x = ops.pre_norm_temporal_attention(src, self.norm1, self.self_attn, self.dropout1.p, self.training)
        else:
            normed, src = ops.layer_norm_residual(src, self.norm1)
            x = self._sa_block(normed, src)
        return self._ff_residual(x)

    def forward_last(self, src: torch.Tensor) -> torch.Tensor:
        """Row ``[-1]`` of ``forward(src)`` ([N, D]) without computing the other rows.  The last position
        attends to every step, so keys and values are still projected for the whole sequence, but the
        query, the output projection, the feed-forward block and both residuals exist for one row per
        agent only -- the encoder's caller reads nothing else of its last layer
        (``models/local_encoder.py:241``)."""
        mha = self.self_attn
        S, N, D = src.shape
        H, hd = mha.num_heads, D // mha.num_heads
        w, b = mha.in_proj_weight, mha.in_proj_bias
        gpu = src.is_cuda

        def lin(x, weight, bias, **kw):
            if gpu:
                return ops.fused_linear(x, weight, bias, **kw)
            y = F.linear(x, weight, bias)
            if kw.get("relu"):
                y = F.relu(y)
            y = F.dropout(y, kw.get("dropout_p", 0.0), kw.get("training", False))
            return y if kw.get("residual") is None else kw["residual"] + y

        if gpu and N > 0 and ops.seg_attn_supported(D, H) and D <= 128 and w.is_contiguous():
            # one query per agent over its own S steps: the segment-attention kernel on the [S, N, 2D] keys and
            # values as they lie (the j-th "edge" of agent n is row j * N + n).  The last-row slices (query input,
            # residual) and the two halves of the fused projection stay inside two nodes of the graph.
            xn, last = ops.layer_norm_last(src, self.norm1, N)
            q, kv = ops.last_row_qkv(xn, w, b, N)
            o = ops.seg_attn(q, kv, _step_rowptr(N, S, src.device), H, mha.dropout, self.training, interleaved=True,
                             tag=type(self).__name__)
            x = lin(o, mha.out_proj.weight, mha.out_proj.bias, dropout_p=self.dropout1.p, training=self.training,
                    residual=last)
            return self._ff_residual(x)
        xn = ops.layer_norm(src, self.norm1)
        if gpu and N > 0 and ops.seg_attn_supported(D, H):
            q = lin(xn[-1], w[:D], b[:D])
            kv = lin(xn, w[D:], b[D:]).view(S * N, 2 * D)
            o = ops.seg_attn(q, kv, _step_rowptr(N, S, src.device), H, mha.dropout, self.training, interleaved=True,
                             tag=type(self).__name__)
        else:
            q = lin(xn[-1], w[:D], b[:D]).view(1, N, H, hd)
            kv = lin(xn, w[D:], b[D:]).view(S, N, 2, H, hd)
            scores = (kv[:, :, 0] * q).sum(dim=-1) * (hd ** -0.5)             # [S, N, H]
            prob = F.dropout(torch.softmax(scores, dim=0), mha.dropout, self.training)
            o = (prob.unsqueeze(-1) * kv[:, :, 1]).sum(dim=0).reshape(N, D)
        x = lin(o, mha.out_proj.weight, mha.out_proj.bias, dropout_p=self.dropout1.p, training=self.training,
                residual=src[-1])
        return self._ff_residual(x)


_step_rowptrs: Dict[Tuple[int, int, str], torch.Tensor] = {}


def _step_rowptr(n: int, steps: int, device: torch.device) -> torch.Tensor:
    """``[0, steps, 2 steps, ..]`` (int32, n + 1 entries): every target owns ``steps`` rows."""
    key = (n, steps, str(device))
    t = _step_rowptrs.get(key)
    if t is None:
        if len(_step_rowptrs) > 64:
            _step_rowptrs.clear()
        t = _step_rowptrs[key] = (torch.arange(n + 1, dtype=torch.int64) * steps).to(torch.int32).to(device)
    return t


class _EncoderStack(nn.Module):
    """``layers`` + final ``norm`` under the key names of ``nn.TransformerEncoder``."""

    def __init__(self, embed_dim: int, num_heads: int, num_layers: int, dropout: float) -> None:
        super().__init__()
        self.layers = nn.ModuleList([TemporalEncoderLayer(embed_dim, num_heads, dropout) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for layer in self.layers:
            x = layer(x)
        return ops.layer_norm(x, self.norm)

    def forward_last(self, x: torch.Tensor) -> torch.Tensor:
        """``forward(x)[-1]``: the last layer and the final norm run for the last position only."""
        for layer in self.layers[:-1]:
            x = layer(x)
        return ops.layer_norm(self.layers[-1].forward_last(x), self.norm)


class TemporalEncoder(nn.Module):
    """Causal transformer over the history with a trailing class token (``models/local_encoder.py:205-241``)."""

    def __init__(self, historical_steps: int, embed_dim: int, num_heads: int = 8, num_layers: int = 4,
                 dropout: float = 0.1) -> None:
        super().__init__()
        self.transformer_encoder = _EncoderStack(embed_dim, num_heads, num_layers, dropout)
        self.padding_token = nn.Parameter(torch.empty(historical_steps, 1, embed_dim))
        self.cls_token = nn.Parameter(torch.empty(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.empty(historical_steps + 1, 1, embed_dim))
        self.register_buffer("attn_mask", self.generate_square_subsequent_mask(historical_steps + 1))
        for token in (self.padding_token, self.cls_token, self.pos_embed):
            nn.init.normal_(token, mean=0.0, std=0.02)
        self.apply(init_weights)

    def forward(self, x: torch.Tensor, padding_mask: torch.Tensor, rows: Optional[ops.RowSubset] = None
                ) -> torch.Tensor:
        """x [T, N, D]; or, with ``rows`` (the unpadded (step, agent) nodes, packed), x [rows.capacity, D]."""
        steps = self.padding_token.shape[0]
        if x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 4 == 0 and x.shape[-1] <= 128 and steps + 1 <= 32 \
                and x.numel() > 0:
            # tokens, packed rows and positional rows in one launch
            x = ops.temporal_input(x, rows, padding_mask, self.padding_token.squeeze(1), self.cls_token.view(-1),
                                   self.pos_embed.squeeze(1))
        elif rows is not None:
            x = ops.expand_rows_with_tokens(x, rows, padding_mask.t(), self.padding_token.squeeze(1))
            x = ops.append_token_add_steps(x, self.cls_token.view(-1), self.pos_embed.squeeze(1))
        elif ops.step_tokens_supported(x, x.shape[0] + 1):
            x = ops.masked_step_tokens(x, padding_mask.t(), self.padding_token.squeeze(1))
            x = ops.append_token_add_steps(x, self.cls_token.view(-1), self.pos_embed.squeeze(1))
        else:
            x = torch.where(padding_mask.t().unsqueeze(-1), self.padding_token, x)
            x = torch.cat((x, self.cls_token.expand(-1, x.shape[1], -1)), dim=0) + self.pos_embed
        return self.transformer_encoder.forward_last(x)

    @staticmethod
    def generate_square_subsequent_mask(seq_len: int) -> torch.Tensor:
        return torch.triu(torch.full((seq_len, seq_len), float("-inf")), diagonal=1)


class LocalEncoder(nn.Module):
    """``models/local_encoder.py:22-84`` with device-side edge construction and batched timesteps."""

    def __init__(self, historical_steps: int, node_dim: int, edge_dim: int, embed_dim: int, num_heads: int = 8,
                 dropout: float = 0.1, num_temporal_layers: int = 4, local_radius: float = 50,
                 parallel: bool = False) -> None:
        super().__init__()
        self.historical_steps = historical_steps
        self.parallel = parallel
        self.local_radius = local_radius
        self.drop_edge = DistanceDropEdge(local_radius)
        self.aa_encoder = AAEncoder(historical_steps=historical_steps, node_dim=node_dim, edge_dim=edge_dim,
                                    embed_dim=embed_dim, num_heads=num_heads, dropout=dropout, parallel=parallel)
        self.temporal_encoder = TemporalEncoder(historical_steps=historical_steps, embed_dim=embed_dim,
                                                num_heads=num_heads, dropout=dropout, num_layers=num_temporal_layers)
        self.al_encoder = ALEncoder(node_dim=node_dim, edge_dim=edge_dim, embed_dim=embed_dim, num_heads=num_heads,
                                    dropout=dropout)

    def forward(self, data) -> torch.Tensor:
        T = self.historical_steps
        n = int(data.num_nodes)
        plan = data.get("edge_plan")
        if plan is None:
            plan = build_edge_plan(data, T, self.local_radius)
            data["edge_plan"] = plan
        prep = data.get("_prep")
        x_all = prep["x_steps"] if prep is not None else data["x"][:, :T].transpose(0, 1).reshape(T * n, -1)
        rows = plan.aa_rows
        if rows is not None and not (x_all.is_cuda and self.aa_encoder.embed_dim <= 128 and T + 1 <= 32):
            rows = None
        out = self.aa_encoder(x=x_all, t=None, edge_index=None, edge_attr=None, bos_mask=data["bos_mask"],
                              rotate_mat=data["rotate_mat"], edges=plan.agent_agent, rows=rows,
                              x_rotated=prep["x_rot"] if prep is not None else None)
        if rows is None:
            out = out.view(T, n, -1)
        out = self.temporal_encoder(x=out, padding_mask=data["padding_mask"][:, :T], rows=rows)
        out = self.al_encoder(x=(data["lane_vectors"], out), edge_index=None, edge_attr=None,
                              is_intersections=data["is_intersections"], turn_directions=data["turn_directions"],
                              traffic_controls=data["traffic_controls"], rotate_mat=data["rotate_mat"],
                              edges=plan.lane_actor, lane_code=prep["lane_code"] if prep is not None else None)
        return out
