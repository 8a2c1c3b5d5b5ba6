"""Functional CPU restatement of the HiVT-GAN hot path (TEST INFRASTRUCTURE ONLY).

Every function takes the reference's parameters as a flat ``{state_dict key: tensor}`` mapping
plus a key prefix, and restates one reference function with plain torch ops.  Autograd through
these functions yields the reference gradients (make the mapping's tensors require grad).

Parity status: the loss / critic functions are pinned by vectors generated from the reference's
own code (see ``oracle/__init__.py``); the edge-attention functions are **parity unpinned**
because their arithmetic lives in torch_geometric, which is absent (un-vendored, un-pinned).

PyG semantics restated here (public behaviour of torch_geometric >= 2.0):

* ``subgraph(mask, ei)``: keep edge e iff mask[ei[0,e]] and mask[ei[1,e]]; order preserved.
* ``propagate`` with flow source_to_target: j = ei[0] (source), i = ei[1] (target);
  ``foo_j`` / ``foo_i`` are ``foo.index_select(0, ei[0])`` / ``(0, ei[1])``; for a 2-tuple ``foo``
  ``_j`` reads element 0 and ``_i`` element 1; aggregation is a sum-scatter over ei[1] into
  ``size_i`` rows (zero rows for targets without incoming edges); ``update`` receives it.
* ``softmax(src, index, ptr=None, num_nodes)``: per target segment and per column,
  ``exp(src - max.detach()) / (sum + 1e-16)``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _drop(x: torch.Tensor, p: float, training: bool) -> torch.Tensor:
    return F.dropout(x, p, training) if (training and p > 0.0) else x


def subgraph(subset: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """torch_geometric.utils.subgraph with a boolean subset (local_encoder.py:59, global_interactor.py:48)."""
    keep = subset[edge_index[0]] & subset[edge_index[1]]
    return edge_index[:, keep]


def distance_drop_edge(edge_index: torch.Tensor, edge_attr: torch.Tensor,
                       max_distance: Optional[float]) -> Tuple[torch.Tensor, torch.Tensor]:
    """utils.py:69-78: keep edges with ||edge_attr||_2 strictly below the radius, order preserved."""
    if max_distance is None:
        return edge_index, edge_attr
    keep = torch.norm(edge_attr, p=2, dim=-1) < max_distance
    return torch.stack([edge_index[0][keep], edge_index[1][keep]], dim=0), edge_attr[keep]


def segment_softmax(alpha: torch.Tensor, index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """torch_geometric.utils.softmax(src, index, None, num_nodes) for src of shape [E, H]."""
    idx = index.unsqueeze(-1).expand_as(alpha)
    mx = torch.zeros(num_nodes, alpha.shape[1], dtype=alpha.dtype).scatter_reduce(
        0, idx, alpha.detach(), "amax", include_self=False)
    ex = (alpha - mx.index_select(0, index)).exp()
    den = torch.zeros(num_nodes, alpha.shape[1], dtype=alpha.dtype).scatter_add(0, idx, ex) + 1e-16
    return ex / den.index_select(0, index)


def scatter_sum(msg: torch.Tensor, index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """aggr='add' of MessagePassing: rows of ``msg`` added into ``num_nodes`` zero rows in edge order."""
    out = torch.zeros((num_nodes,) + tuple(msg.shape[1:]), dtype=msg.dtype)
    return out.index_add(0, index, msg)


def _edge_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, index: torch.Tensor,
                    num_nodes: int, num_heads: int, p: float, training: bool) -> torch.Tensor:
    """Shared tail of the three ``message`` functions followed by the sum aggregation.

    local_encoder.py:174-181,364-371 and global_interactor.py:120-124; returns [num_nodes, D]."""
    e, d_model = key.shape
    d = d_model // num_heads
    q = query.view(-1, num_heads, d)
    k = key.view(-1, num_heads, d)
    v = value.view(-1, num_heads, d)
    alpha = (q * k).sum(dim=-1) / (d ** 0.5)
    alpha = segment_softmax(alpha, index, num_nodes)
    alpha = _drop(alpha, p, training)
    msg = v * alpha.unsqueeze(-1)
    return scatter_sum(msg, index, num_nodes).view(num_nodes, d_model)


def _gated_update(sd: SD, p: str, inputs: torch.Tensor, center: torch.Tensor) -> torch.Tensor:
    """``update`` of all three modules (local_encoder.py:183-188,373-379; global_interactor.py:126-131)."""
    gate = torch.sigmoid(_lin(sd, p + ".lin_ih", inputs) + _lin(sd, p + ".lin_hh", center))
    return inputs + gate * (_lin(sd, p + ".lin_self", center) - inputs)


def _ff_block(sd: SD, p: str, x: torch.Tensor, drop: float, training: bool) -> torch.Tensor:
    h = _drop(F.relu(_lin(sd, p + ".mlp.0", x)), drop, training)
    return _drop(_lin(sd, p + ".mlp.3", h), drop, training)


# --------------------------------------------------------------------------------------
# embeddings (models/embedding.py)
# --------------------------------------------------------------------------------------
def single_input_embedding(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """embedding.py:15-27."""
    h = F.relu(_ln(sd, p + ".embed.1", _lin(sd, p + ".embed.0", x)))
    h = F.relu(_ln(sd, p + ".embed.4", _lin(sd, p + ".embed.3", h)))
    return _ln(sd, p + ".embed.7", _lin(sd, p + ".embed.6", h))


def multiple_input_embedding(sd: SD, p: str, continuous: List[torch.Tensor],
                             categorical: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """embedding.py:49-57."""
    branches = []
    for i, c in enumerate(continuous):
        h = F.relu(_ln(sd, f"{p}.module_list.{i}.1", _lin(sd, f"{p}.module_list.{i}.0", c)))
        branches.append(_lin(sd, f"{p}.module_list.{i}.3", h))
    out = torch.stack(branches).sum(dim=0)
    if categorical is not None:
        out = out + torch.stack(categorical).sum(dim=0)
    h = F.relu(_ln(sd, p + ".aggr_embed.0", out))
    return _ln(sd, p + ".aggr_embed.3", _lin(sd, p + ".aggr_embed.2", h))


# --------------------------------------------------------------------------------------
# AAEncoder (models/local_encoder.py:87-202)
# --------------------------------------------------------------------------------------
def aa_encoder(sd: SD, p: str, x: torch.Tensor, t: Optional[int], edge_index: torch.Tensor,
               edge_attr: torch.Tensor, bos_mask: torch.Tensor, rotate_mat: Optional[torch.Tensor],
               historical_steps: int, num_heads: int = 8, parallel: bool = False,
               dropout: float = 0.0, training: bool = False) -> torch.Tensor:
    T = historical_steps
    if parallel:  # local_encoder.py:135-144
        xv = x.view(T, x.shape[0] // T, -1)
        if rotate_mat is None:
            c = single_input_embedding(sd, p + ".center_embed", xv)
        else:
            c = single_input_embedding(
                sd, p + ".center_embed",
                torch.matmul(xv.unsqueeze(-2), rotate_mat.expand(T, *rotate_mat.shape)).squeeze(-2))
        c = torch.where(bos_mask.t().unsqueeze(-1), sd[p + ".bos_token"].unsqueeze(-2), c).reshape(x.shape[0], -1)
    else:  # local_encoder.py:146-150
        if rotate_mat is None:
            c = single_input_embedding(sd, p + ".center_embed", x)
        else:
            c = single_input_embedding(sd, p + ".center_embed",
                                       torch.bmm(x.unsqueeze(-2), rotate_mat).squeeze(-2))
        c = torch.where(bos_mask.unsqueeze(-1), sd[p + ".bos_token"][t], c)

    # _mha_block(norm1(c), ...) -> propagate -> message / aggregate / update (local_encoder.py:151,156-199)
    cn = _ln(sd, p + ".norm1", c)
    src, dst = edge_index[0], edge_index[1]
    x_j = x.index_select(0, src)
    c_i = cn.index_select(0, dst)
    if rotate_mat is None:
        nbr = multiple_input_embedding(sd, p + ".nbr_embed", [x_j, edge_attr])
    else:
        rot = (rotate_mat.repeat(T, 1, 1) if parallel else rotate_mat)[dst]
        nbr = multiple_input_embedding(sd, p + ".nbr_embed",
                                       [torch.bmm(x_j.unsqueeze(-2), rot).squeeze(-2),
                                        torch.bmm(edge_attr.unsqueeze(-2), rot).squeeze(-2)])
    agg = _edge_attention(_lin(sd, p + ".lin_q", c_i), _lin(sd, p + ".lin_k", nbr), _lin(sd, p + ".lin_v", nbr),
                          dst, cn.shape[0], num_heads, dropout, training)
    upd = _gated_update(sd, p, agg, cn)
    c = c + _drop(_lin(sd, p + ".out_proj", upd), dropout, training)
    c = c + _ff_block(sd, p, _ln(sd, p + ".norm2", c), dropout, training)
    return c


# --------------------------------------------------------------------------------------
# ALEncoder (models/local_encoder.py:283-397)
# --------------------------------------------------------------------------------------
def al_encoder(sd: SD, p: str, x_lane: torch.Tensor, x_actor: torch.Tensor, edge_index: torch.Tensor,
               edge_attr: torch.Tensor, is_intersections: torch.Tensor, turn_directions: torch.Tensor,
               traffic_controls: torch.Tensor, rotate_mat: Optional[torch.Tensor], num_heads: int = 8,
               dropout: float = 0.0, training: bool = False) -> torch.Tensor:
    ii, td, tc = is_intersections.long(), turn_directions.long(), traffic_controls.long()
    xn = _ln(sd, p + ".norm1", x_actor)
    src, dst = edge_index[0], edge_index[1]
    x_j = x_lane.index_select(0, src)
    x_i = xn.index_select(0, dst)
    cat = [sd[p + ".is_intersection_embed"][ii.index_select(0, src)],
           sd[p + ".turn_direction_embed"][td.index_select(0, src)],
           sd[p + ".traffic_control_embed"][tc.index_select(0, src)]]
    if rotate_mat is None:
        lane = multiple_input_embedding(sd, p + ".lane_embed", [x_j, edge_attr], cat)
    else:
        rot = rotate_mat[dst]
        lane = multiple_input_embedding(sd, p + ".lane_embed",
                                        [torch.bmm(x_j.unsqueeze(-2), rot).squeeze(-2),
                                         torch.bmm(edge_attr.unsqueeze(-2), rot).squeeze(-2)], cat)
    agg = _edge_attention(_lin(sd, p + ".lin_q", x_i), _lin(sd, p + ".lin_k", lane), _lin(sd, p + ".lin_v", lane),
                          dst, x_actor.shape[0], num_heads, dropout, training)
    upd = _gated_update(sd, p, agg, xn)  # update() reads x[1] = norm1(x_actor) (local_encoder.py:376)
    x_actor = x_actor + _drop(_lin(sd, p + ".out_proj", upd), dropout, training)
    x_actor = x_actor + _ff_block(sd, p, _ln(sd, p + ".norm2", x_actor), dropout, training)
    return x_actor


# --------------------------------------------------------------------------------------
# GlobalInteractorLayer / GlobalInteractor (models/global_interactor.py)
# --------------------------------------------------------------------------------------
def global_interactor_layer(sd: SD, p: str, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor,
                            num_heads: int = 8, dropout: float = 0.0, training: bool = False) -> torch.Tensor:
    """global_interactor.py:99-142."""
    xn = _ln(sd, p + ".norm1", x)
    src, dst = edge_index[0], edge_index[1]
    x_i, x_j = xn.index_select(0, dst), xn.index_select(0, src)
    query = _lin(sd, p + ".lin_q_node", x_i)
    key = _lin(sd, p + ".lin_k_node", x_j) + _lin(sd, p + ".lin_k_edge", edge_attr)
    value = _lin(sd, p + ".lin_v_node", x_j) + _lin(sd, p + ".lin_v_edge", edge_attr)
    agg = _edge_attention(query, key, value, dst, x.shape[0], num_heads, dropout, training)
    upd = _gated_update(sd, p, agg, xn)
    x = x + _drop(_lin(sd, p + ".out_proj", upd), dropout, training)
    x = x + _ff_block(sd, p, _ln(sd, p + ".norm2", x), dropout, training)
    return x


def global_interactor(sd: SD, p: str, data: dict, local_embed: torch.Tensor, historical_steps: int,
                      num_modes: int, num_layers: int, num_heads: int = 8, dropout: float = 0.0,
                      training: bool = False) -> torch.Tensor:
    """global_interactor.py:45-65 (rotate=True branch when ``data['rotate_mat']`` is not None)."""
    t = historical_steps - 1
    edge_index = subgraph(~data["padding_mask"][:, t], data["edge_index"])
    rel_pos = data["positions"][edge_index[0], t] - data["positions"][edge_index[1], t]
    if data.get("rotate_mat") is None:
        rel_embed = single_input_embedding(sd, p + ".rel_embed", rel_pos)
    else:
        rel_pos = torch.bmm(rel_pos.unsqueeze(-2), data["rotate_mat"][edge_index[1]]).squeeze(-2)
        rel_theta = data["rotate_angles"][edge_index[0]] - data["rotate_angles"][edge_index[1]]
        rel_embed = multiple_input_embedding(
            sd, p + ".rel_embed",
            [rel_pos, torch.cat((torch.cos(rel_theta).unsqueeze(-1), torch.sin(rel_theta).unsqueeze(-1)), dim=-1)])
    x = local_embed
    for i in range(num_layers):
        x = global_interactor_layer(sd, f"{p}.global_interactor_layers.{i}", x, edge_index, rel_embed,
                                    num_heads, dropout, training)
    x = _ln(sd, p + ".norm", x)
    d_model = x.shape[-1]
    x = _lin(sd, p + ".multihead_proj", x).view(-1, num_modes, d_model)
    return x.transpose(0, 1)


# --------------------------------------------------------------------------------------
# TemporalEncoder (models/local_encoder.py:205-280)
# --------------------------------------------------------------------------------------
def _multihead_self_attention(sd: SD, p: str, x: torch.Tensor, attn_mask: torch.Tensor, num_heads: int,
                              dropout: float, training: bool) -> torch.Tensor:
    """nn.MultiheadAttention(x, x, x, attn_mask) for x of shape [S, N, D] (batch_first=False)."""
    S, N, D = x.shape
    d = D // num_heads
    qkv = F.linear(x, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.reshape(S, N * num_heads, d).transpose(0, 1)
    k = k.reshape(S, N * num_heads, d).transpose(0, 1)
    v = v.reshape(S, N * num_heads, d).transpose(0, 1)
    scores = torch.baddbmm(attn_mask, q * (1.0 / math.sqrt(d)), k.transpose(-2, -1))
    w = _drop(F.softmax(scores, dim=-1), dropout, training)
    out = torch.bmm(w, v).transpose(0, 1).reshape(S, N, D)
    return F.linear(out, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def temporal_encoder(sd: SD, p: str, x: torch.Tensor, padding_mask: torch.Tensor, num_layers: int,
                     num_heads: int = 8, dropout: float = 0.0, training: bool = False) -> torch.Tensor:
    """local_encoder.py:227-235 with pre-norm layers (local_encoder.py:260-280) and a final LayerNorm."""
    x = torch.where(padding_mask.t().unsqueeze(-1), sd[p + ".padding_token"], x)
    x = torch.cat((x, sd[p + ".cls_token"].expand(-1, x.shape[1], -1)), dim=0)
    x = x + sd[p + ".pos_embed"]
    S = x.shape[0]
    mask = torch.triu(torch.full((S, S), float("-inf"), dtype=x.dtype), diagonal=1)
    for i in range(num_layers):
        lp = f"{p}.transformer_encoder.layers.{i}"
        h = _multihead_self_attention(sd, lp + ".self_attn", _ln(sd, lp + ".norm1", x), mask, num_heads,
                                      dropout, training)
        x = x + _drop(h, dropout, training)
        h = _lin(sd, lp + ".linear2", _drop(F.relu(_lin(sd, lp + ".linear1", _ln(sd, lp + ".norm2", x))),
                                            dropout, training))
        x = x + _drop(h, dropout, training)
    x = _ln(sd, p + ".transformer_encoder.norm", x)
    return x[-1]


# --------------------------------------------------------------------------------------
# LocalEncoder (models/local_encoder.py:57-84)
# --------------------------------------------------------------------------------------
def temporal_edges(data: dict, t: int, local_radius: Optional[float]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Edges of timestep t: subgraph by validity, relative positions, radius filter
    (local_encoder.py:59-61,75)."""
    ei = subgraph(~data["padding_mask"][:, t], data["edge_index"])
    attr = data["positions"][ei[0], t] - data["positions"][ei[1], t]
    return distance_drop_edge(ei, attr, local_radius)


def local_encoder(sd: SD, p: str, data: dict, historical_steps: int, num_temporal_layers: int,
                  local_radius: Optional[float], num_heads: int = 8, parallel: bool = False, dropout: float = 0.0,
                  training: bool = False) -> torch.Tensor:
    T = historical_steps
    n = data["x"].shape[0]
    if parallel:  # local_encoder.py:62-71: snapshots batched with node id t*N + n
        eis, eas = [], []
        for t in range(T):
            ei, ea = temporal_edges(data, t, local_radius)
            eis.append(ei + t * n)
            eas.append(ea)
        out = aa_encoder(sd, p + ".aa_encoder", data["x"].transpose(0, 1).reshape(T * n, -1), None,
                         torch.cat(eis, dim=1), torch.cat(eas, dim=0), data["bos_mask"], data["rotate_mat"], T,
                         num_heads, True, dropout, training)
        out = out.view(T, n, -1)
    else:  # local_encoder.py:73-78
        outs = []
        for t in range(T):
            ei, ea = temporal_edges(data, t, local_radius)
            outs.append(aa_encoder(sd, p + ".aa_encoder", data["x"][:, t], t, ei, ea, data["bos_mask"][:, t],
                                   data["rotate_mat"], T, num_heads, False, dropout, training))
        out = torch.stack(outs)
    out = temporal_encoder(sd, p + ".temporal_encoder", out, data["padding_mask"][:, :T], num_temporal_layers,
                           num_heads, dropout, training)
    ei, ea = distance_drop_edge(data["lane_actor_index"], data["lane_actor_vectors"], local_radius)
    return al_encoder(sd, p + ".al_encoder", data["lane_vectors"], out, ei, ea, data["is_intersections"],
                      data["turn_directions"], data["traffic_controls"], data["rotate_mat"], num_heads, dropout,
                      training)


# --------------------------------------------------------------------------------------
# MLPDecoder (models/decoder.py:116-139)
# --------------------------------------------------------------------------------------
def mlp_decoder(sd: SD, p: str, local_embed: torch.Tensor, global_embed: torch.Tensor, future_steps: int,
                num_modes: int) -> Tuple[torch.Tensor, torch.Tensor]:
    loc_exp = local_embed.expand(num_modes, *local_embed.shape)
    h = torch.cat((loc_exp, global_embed), dim=-1)
    h = F.relu(_ln(sd, p + ".pi.1", _lin(sd, p + ".pi.0", h)))
    h = F.relu(_ln(sd, p + ".pi.4", _lin(sd, p + ".pi.3", h)))
    pi = _lin(sd, p + ".pi.6", h).squeeze(-1).t()
    out = F.relu(_ln(sd, p + ".aggr_embed.1", _lin(sd, p + ".aggr_embed.0", torch.cat((global_embed, loc_exp), dim=-1))))
    loc = _lin(sd, p + ".loc.3", F.relu(_ln(sd, p + ".loc.1", _lin(sd, p + ".loc.0", out))))
    loc = loc.view(num_modes, -1, future_steps, 2)
    scale = _lin(sd, p + ".scale.3", F.relu(_ln(sd, p + ".scale.1", _lin(sd, p + ".scale.0", out))))
    scale = F.elu(scale, alpha=1.0) + 1.0
    scale = torch.clamp(scale.view(num_modes, -1, future_steps, 2), min=0.05)
    return torch.cat((loc, scale), dim=-1), pi


# --------------------------------------------------------------------------------------
# HiVT.forward (models/hivt.py:151-169)
# --------------------------------------------------------------------------------------
def hivt_forward(sd: SD, data: dict, cfg: dict, training: bool = False
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Returns (y_hat[F,N,H,4], pi[N,F], y_rot[N,H,2]); ``data`` is not modified (the reference
    rotates ``data.y`` in place, hivt.py:160-161 -- the rotated copy is returned instead)."""
    d = dict(data)
    if cfg.get("rotate", True):
        ang = d["rotate_angles"]
        s, c = torch.sin(ang), torch.cos(ang)
        rot = torch.stack([torch.stack([c, -s], dim=-1), torch.stack([s, c], dim=-1)], dim=-2)
        y_rot = torch.bmm(d["y"], rot)
        d["rotate_mat"] = rot
    else:
        y_rot = d["y"]
        d["rotate_mat"] = None
    T, H, drop = cfg["historical_steps"], cfg.get("num_heads", 8), cfg.get("dropout", 0.0)
    local = local_encoder(sd, "local_encoder", d, T, cfg.get("num_temporal_layers", 4),
                          cfg.get("local_radius", 50.0), H, cfg.get("parallel", False), drop, training)
    glob = global_interactor(sd, "global_interactor", d, local, T, cfg["num_modes"],
                             cfg.get("num_global_layers", 3), H, drop, training)
    y_hat, pi = mlp_decoder(sd, "decoder", local, glob, cfg["future_steps"], cfg["num_modes"])
    return y_hat, pi, y_rot


# --------------------------------------------------------------------------------------
# losses (losses/*.py, models/hivt.py:174-256)
# --------------------------------------------------------------------------------------
def laplace_nll(pred: torch.Tensor, target: torch.Tensor, eps: float = 1e-6, reduction: str = "mean") -> torch.Tensor:
    """losses/laplace_nll_loss.py:12-36 (channels 2:4 are treated as a log-scale)."""
    loc, log_scale = pred.chunk(2, dim=-1)
    log_scale = torch.clamp(log_scale, min=-10.0, max=10.0)
    nll = log_scale + math.log(2.0) + torch.abs(target - loc) / (torch.exp(log_scale) + eps)
    if reduction == "mean":
        return nll.mean()
    if reduction == "sum":
        return nll.sum()
    if reduction == "none":
        return nll
    raise ValueError(f"{reduction} is not a valid value for reduction")


def soft_target_cross_entropy(pred: torch.Tensor, target: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
    """losses/soft_target_cross_entropy_loss.py:12-23."""
    ce = torch.sum(-target * F.log_softmax(pred, dim=-1), dim=-1)
    if reduction == "mean":
        return ce.mean()
    if reduction == "sum":
        return ce.sum()
    if reduction == "none":
        return ce
    raise ValueError(f"{reduction} is not a valid value for reduction")


def supervised_losses(y_hat: torch.Tensor, pi: torch.Tensor, y: torch.Tensor, padding_mask: torch.Tensor,
                      historical_steps: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """hivt.py:174-209; the empty-mask branch returns three values (zero, zero, zeros)."""
    reg_mask = ~padding_mask[:, historical_steps:]
    n = y.shape[0]
    if reg_mask.sum() == 0:
        zero = y_hat.sum() * 0.0
        return zero, zero, torch.zeros(n, dtype=torch.long)
    valid_steps = reg_mask.sum(dim=-1)
    cls_mask = valid_steps > 0
    l2 = (torch.norm(y_hat[:, :, :, :2] - y, p=2, dim=-1) * reg_mask).sum(dim=-1)
    best = l2.argmin(dim=0)
    y_best = y_hat[best, torch.arange(n)]
    reg = laplace_nll(y_best[reg_mask], y[reg_mask])
    avg = -l2[:, cls_mask] / (valid_steps[cls_mask] + 1e-6)
    soft = F.softmax(avg, dim=0).t().detach()
    cls = soft_target_cross_entropy(pi[cls_mask], soft)
    return reg, cls, best


def build_real_fake(y_hat: torch.Tensor, best_mode: torch.Tensor, y: torch.Tensor, padding_mask: torch.Tensor,
                    historical_steps: int, short_horizon: int, mid_horizon: int):
    """hivt.py:215-256."""
    n = y.shape[0]
    fake = y_hat[best_mode, torch.arange(n), :, :2]
    keep = (~padding_mask[:, historical_steps:]).any(dim=1)
    real, fake = y[keep], fake[keep]
    return ({"short": real[:, :short_horizon], "mid": real[:, :mid_horizon], "long": real},
            {"short": fake[:, :short_horizon], "mid": fake[:, :mid_horizon], "long": fake}, keep)


def critic_forward(sd: SD, p: str, traj: torch.Tensor, horizon: int, num_layers: int, dropout: float = 0.0,
                   training: bool = True) -> torch.Tensor:
    """BaseTrajectoryCritic.forward (models/critics/base_critic.py:36-44) on ``traj[:, :horizon]``
    (short/mid/long_scale_critic.py:24-29) with the legacy hook-based spectral norm: in training
    mode every forward runs one power iteration that overwrites ``weight_u`` / ``weight_v`` in
    ``sd`` (no grad), then divides ``weight_orig`` by sigma = u . (W v)."""

    def sn_linear(name: str, x: torch.Tensor) -> torch.Tensor:
        w = sd[name + ".weight_orig"]
        u, v = sd[name + ".weight_u"], sd[name + ".weight_v"]
        if training:
            with torch.no_grad():
                v.copy_(F.normalize(torch.mv(w.t(), u), dim=0, eps=1e-12))
                u.copy_(F.normalize(torch.mv(w, v), dim=0, eps=1e-12))
            u, v = u.clone(), v.clone()
        sigma = torch.dot(u, torch.mv(w, v))
        return F.linear(x, w / sigma, sd[name + ".bias"])

    traj = traj[:, :horizon]
    h = traj.reshape(traj.shape[0], -1)
    for i in range(num_layers):
        h = _drop(F.leaky_relu(sn_linear(f"{p}.mlp.{3 * i}", h), 0.2), dropout, training)
    return sn_linear(p + ".out", h).squeeze(-1)


def r1_penalty(d_out: torch.Tensor, real: torch.Tensor) -> torch.Tensor:
    """losses/r1_regularization.py:27-36."""
    g = torch.autograd.grad(outputs=d_out.sum(), inputs=real, create_graph=True, retain_graph=True,
                            only_inputs=True)[0]
    return g.pow(2).reshape(g.size(0), -1).sum(dim=1).mean()


CRITIC_LAYERS = {"short": 2, "mid": 3, "long": 4}  # short/mid/long_scale_critic.py:18-22
CRITIC_PREFIX = {"short": "D_short", "mid": "D_mid", "long": "D_long"}  # hivt.py:135-137


def discriminator_loss(sd: SD, real_trajs: dict, fake_trajs: dict, horizons: dict, lambda_r1: float,
                       dropout: float = 0.0, training: bool = True):
    """losses/adv_losses.py:20-58 over the three critics in dict order short, mid, long (hivt.py:139-143)."""
    total, logs = 0.0, {}
    for name in ("short", "mid", "long"):
        real = real_trajs[name].requires_grad_(True)
        fake = fake_trajs[name].detach()
        d_real = critic_forward(sd, CRITIC_PREFIX[name], real, horizons[name], CRITIC_LAYERS[name], dropout, training)
        d_fake = critic_forward(sd, CRITIC_PREFIX[name], fake, horizons[name], CRITIC_LAYERS[name], dropout, training)
        d_loss = F.relu(1.0 - d_real).mean() + F.relu(1.0 + d_fake).mean()
        r1 = r1_penalty(d_real, real)
        d_loss = d_loss + lambda_r1 * r1
        total = total + d_loss
        logs[f"d_loss_{name}"] = d_loss.detach()
        logs[f"r1_{name}"] = r1.detach()
    return total, logs


def generator_adv_loss(sd: SD, fake_trajs: dict, horizons: dict, lambda_adv: float, dropout: float = 0.0,
                       training: bool = True):
    """losses/adv_losses.py:74-99."""
    total, logs = 0.0, {}
    for name in ("short", "mid", "long"):
        g = -critic_forward(sd, CRITIC_PREFIX[name], fake_trajs[name], horizons[name], CRITIC_LAYERS[name],
                            dropout, training).mean()
        total = total + g
        logs[f"g_loss_{name}"] = g.detach()
    return lambda_adv * total, logs


# --------------------------------------------------------------------------------------
# One G + D + R1 training step, gradients only (hivt.py:290-327 without the optimizer updates)
# --------------------------------------------------------------------------------------
def gan_step_grads(gen_sd: SD, critic_sd: SD, data: dict, cfg: dict, training: bool = True) -> dict:
    """Restates ``training_step`` Case B up to the two backward calls and returns losses and
    gradients.  ``gen_sd`` / ``critic_sd`` tensors must require grad (spectral-norm buffers excepted).
    The critic update between the two backwards (``opt_d.step()``, hivt.py:314) is left to the
    caller through ``cfg['critic_update']`` (a callable run after the D gradients are taken)."""
    T = cfg["historical_steps"]
    horizons = {"short": cfg.get("short_horizon", 10), "mid": cfg.get("mid_horizon", 20), "long": cfg["future_steps"]}
    y_hat, pi, y_rot = hivt_forward(gen_sd, data, cfg, training)
    reg, cls, best = supervised_losses(y_hat, pi, y_rot, data["padding_mask"], T)
    real, fake, keep = build_real_fake(y_hat, best, y_rot, data["padding_mask"], T, horizons["short"], horizons["mid"])
    has_valid = real["long"].shape[0] > 0
    drop = cfg.get("critic_dropout", 0.1 if training else 0.0)
    d_params = [v for v in critic_sd.values() if v.requires_grad]
    if has_valid:
        d_loss, d_logs = discriminator_loss(critic_sd, real, fake, horizons, cfg.get("lambda_r1", 1.0), drop, training)
    else:
        d_loss, d_logs = y_hat.sum() * 0.0, {}
    d_grads = torch.autograd.grad(d_loss, d_params, allow_unused=True, retain_graph=True) if has_valid else \
        [None] * len(d_params)
    if cfg.get("critic_update") is not None:
        cfg["critic_update"](dict(zip([k for k, v in critic_sd.items() if v.requires_grad], d_grads)))
    if has_valid:
        g_adv, g_logs = generator_adv_loss(critic_sd, fake, horizons, cfg.get("lambda_adv", 0.1), drop, training)
    else:
        g_adv, g_logs = y_hat.sum() * 0.0, {}
    g_total = reg + cls + g_adv
    g_keys = [k for k, v in gen_sd.items() if v.requires_grad]
    g_grads = torch.autograd.grad(g_total, [gen_sd[k] for k in g_keys], allow_unused=True)
    return {"y_hat": y_hat.detach(), "pi": pi.detach(), "best_mode": best, "reg_loss": reg.detach(),
            "cls_loss": cls.detach(), "d_loss": d_loss.detach(), "g_adv": g_adv.detach(),
            "g_total": g_total.detach(), "d_logs": d_logs, "g_logs": g_logs,
            "d_grads": dict(zip([k for k, v in critic_sd.items() if v.requires_grad], d_grads)),
            "g_grads": dict(zip(g_keys, g_grads))}
