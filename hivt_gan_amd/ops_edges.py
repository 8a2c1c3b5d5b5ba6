"""Edge construction on the device (filters, CSR grouping, all-pairs sets) and the gather + rotate of the per-edge
features: wrappers over ``edges.hip`` and ``gather.hip``."""
from __future__ import annotations

import contextlib
import ctypes
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from ._opsbase import (UnsupportedShapeError, _ld, _need_gpu, _p, _ptr_array, _rows, _rows2, _stream, _unsupported,
                       _workspace)


# --------------------------------------------------------------------------------------------
# Edge construction
# --------------------------------------------------------------------------------------------
def edges_temporal(edge_index: torch.Tensor, padding_mask: torch.Tensor, positions: torch.Tensor, t0: int, T: int,
                   radius: Optional[float]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Validity sub-graph + relative positions + radius filter for timesteps ``t0..t0+T-1`` at once.

    Returns capacity-sized ``src, dst`` (int32, node ids offset by ``(t-t0)*N``), ``attr`` (float [.,2])
    and a device int32 ``count``; only the first ``count`` entries are meaningful.
    """
    _need_gpu(edge_index, padding_mask, positions)
    edge_index = edge_index.contiguous()
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"edge_index must be int64 [2, E], got {edge_index.dtype} {tuple(edge_index.shape)}")
    pm = padding_mask.contiguous()
    pm = pm.view(torch.uint8) if pm.dtype == torch.bool else pm
    pos = positions.contiguous()
    if pos.dtype != torch.float32 or pos.dim() != 3 or pos.shape[2] != 2:
        raise ValueError("positions must be float32 [N, steps, 2]")
    n = pos.shape[0]
    if t0 < 0 or t0 + T > pm.shape[1] or t0 + T > pos.shape[1]:
        raise ValueError("timestep range outside padding_mask / positions")
    e0 = edge_index.shape[1]
    cap = max(e0 * T, 1)
    dev = pos.device
    src = torch.empty(cap, dtype=torch.int32, device=dev)
    dst = torch.empty(cap, dtype=torch.int32, device=dev)
    attr = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    ws_bytes = _lib.query("hivt_edges_workspace_bytes", e0 * T)
    ws = _workspace(ws_bytes, dev)
    _lib.call("hivt_edges_temporal", _p(edge_index), e0, _p(pm), pm.stride(0), _p(pos), pos.stride(0), pos.stride(1),
              n, t0, T, float(radius) if radius is not None else 0.0, _p(src), _p(dst), _p(attr), _p(count), _p(ws),
              ws.numel(), _stream(pos))
    return src, dst, attr, count


def edges_radius(edge_index: torch.Tensor, edge_attr: torch.Tensor, radius: Optional[float]
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Order-preserving radius filter of a COO edge list (lane -> actor edges)."""
    _need_gpu(edge_index, edge_attr)
    edge_index = edge_index.contiguous()
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"edge_index must be int64 [2, E], got {edge_index.dtype} {tuple(edge_index.shape)}")
    attr_in = edge_attr.contiguous()
    if attr_in.dtype != torch.float32 or attr_in.shape != (edge_index.shape[1], 2):
        raise ValueError("edge_attr must be float32 [E, 2]")
    e0 = edge_index.shape[1]
    cap = max(e0, 1)
    dev = attr_in.device
    src = torch.empty(cap, dtype=torch.int32, device=dev)
    dst = torch.empty(cap, dtype=torch.int32, device=dev)
    attr = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    ws = _workspace(_lib.query("hivt_edges_workspace_bytes", e0), dev)
    _lib.call("hivt_edges_radius", _p(edge_index), e0, _p(attr_in), float(radius) if radius is not None else 0.0,
              _p(src), _p(dst), _p(attr), _p(count), _p(ws), ws.numel(), _stream(attr_in))
    return src, dst, attr, count


def csr_build(key: torch.Tensor, num_rows: int, count: Optional[torch.Tensor] = None, order: bool = False):
    """Stable grouping of edge positions by ``key``: returns (rowptr int32 [num_rows+1], perm int32 [E]).
    ``count`` (device int32 scalar) marks the live prefix of a list padded to a fixed capacity.  With ``order`` a
    third result: the rows ranked by length, longest first (int32 [num_rows]; the schedule that ``seg_attn`` and
    the attention blocks take as ``row_order``)."""
    _need_gpu(key)
    if key.dtype != torch.int32 or key.dim() != 1:
        raise ValueError("key must be an int32 vector")
    key = key.contiguous()
    e = key.shape[0]
    dev = key.device
    rowptr = torch.empty(num_rows + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(e, dtype=torch.int32, device=dev)
    ranked = torch.empty(num_rows, dtype=torch.int32, device=dev) if order else None
    ws = _workspace(_lib.query("hivt_csr_workspace_bytes", e, num_rows), dev)
    _lib.call("hivt_csr_build", _p(key), e, _p(count), num_rows, _p(rowptr), _p(perm), _p(ranked), _p(ws), ws.numel(),
              _stream(key))
    return (rowptr, perm, ranked) if order else (rowptr, perm)


def edges_all_pairs_csr(scene_range: torch.Tensor, padding_mask: torch.Tensor, positions: torch.Tensor, t0: int,
                        T: int, radius: Optional[float], capacity: int
                        ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Agent-agent edges of timesteps ``t0..t0+T-1`` for all-pairs candidate sets, produced in CSR order
    without a candidate list or a sort.  Returns ``rowptr`` (int32 [T*N + 1], clamped to ``capacity``),
    ``src``, ``dst`` (int32 [capacity], ids offset by ``(t-t0)*N``), ``attr`` (float [capacity, 2]) and the
    device int32 ``count`` of kept edges before clamping."""
    _need_gpu(scene_range, padding_mask, positions)
    if scene_range.dtype != torch.int32 or scene_range.dim() != 2 or scene_range.shape[1] != 2:
        raise ValueError("scene_range must be int32 [N, 2]")
    sr = scene_range.contiguous()
    pm = padding_mask.contiguous()
    pm = pm.view(torch.uint8) if pm.dtype == torch.bool else pm
    pos = positions.contiguous()
    if pos.dtype != torch.float32 or pos.dim() != 3 or pos.shape[2] != 2:
        raise ValueError("positions must be float32 [N, steps, 2]")
    n = pos.shape[0]
    if sr.shape[0] != n or pm.shape[0] != n:
        raise ValueError("scene_range, padding_mask and positions must describe the same agents")
    if t0 < 0 or t0 + T > pm.shape[1] or t0 + T > pos.shape[1]:
        raise ValueError("timestep range outside padding_mask / positions")
    dev = pos.device
    cap = max(int(capacity), 1)
    rowptr = torch.empty(n * T + 1, dtype=torch.int32, device=dev)
    src = torch.empty(cap, dtype=torch.int32, device=dev)
    dst = torch.empty(cap, dtype=torch.int32, device=dev)
    attr = torch.empty(cap, 2, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    ws = _workspace(_lib.query("hivt_edges_all_pairs_workspace_bytes", n * T), dev)
    _lib.call("hivt_edges_all_pairs_csr", _p(sr), _p(pm), pm.stride(0), _p(pos), pos.stride(0), pos.stride(1), n, t0, T,
              float(radius) if radius is not None else 0.0, int(capacity), _p(rowptr), _p(src), _p(dst), _p(attr),
              _p(count), _p(ws), ws.numel(), _stream(pos))
    return rowptr, src, dst, attr, count


@dataclass
class EdgeSet:
    """One filtered edge list grouped by target node.

    ``coo_*`` are in the order the reference would hold them (filtered, order preserving);
    ``src`` / ``dst`` are the same edges in CSR order (position i of the CSR order is position
    ``perm[i]`` of the COO order).  A set produced directly in CSR order has ``perm`` None and its
    ``coo_*`` lists already grouped by target."""
    num_src: int
    num_dst: int
    coo_src: torch.Tensor
    coo_dst: torch.Tensor
    coo_attr: torch.Tensor
    rowptr: torch.Tensor
    perm: Optional[torch.Tensor]
    src: Optional[torch.Tensor] = None
    dst: Optional[torch.Tensor] = None
    rowptr_by_src: Optional[torch.Tensor] = None
    perm_by_src: Optional[torch.Tensor] = None
    count: Optional[torch.Tensor] = None  # device int32: live edges when the lists are padded to a capacity
    row_order: Optional[torch.Tensor] = None  # int32 [num_dst]: target rows, longest first (a launch schedule)

    @property
    def num_edges(self) -> int:
        return int(self.coo_src.shape[0])

    def edge_index(self) -> torch.Tensor:
        """The filtered COO list as the reference's int64 ``[2, E]`` tensor."""
        return torch.stack([self.coo_src, self.coo_dst]).long()

    def group_by_source(self) -> None:
        """CSR over sources of the CSR-ordered edges (for gradients of source lifts)."""
        if self.rowptr_by_src is None:
            self.rowptr_by_src, self.perm_by_src = csr_build(self.src, self.num_src, self.count)


def make_edge_set(src: torch.Tensor, dst: torch.Tensor, attr: torch.Tensor, num_src: int, num_dst: int,
                  count: Optional[torch.Tensor] = None, order: bool = False) -> EdgeSet:
    """Group an already filtered int32 COO list by target (``count``: live prefix of padded lists; ``order``: also
    rank the target rows by length, for sets whose row lengths are spread widely)."""
    rowptr, perm, *ranked = csr_build(dst, num_dst, count, order)
    return EdgeSet(num_src, num_dst, src, dst, attr, rowptr, perm, count=count, row_order=ranked[0] if order else None)


# --------------------------------------------------------------------------------------------
# Kernel (a): gather + rotate
# --------------------------------------------------------------------------------------------
class _EdgeGatherRotate(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x_src, attr, rot, src, dst, perm, rot_mod, count):
        _need_gpu(x_src, attr, src, dst)
        x2 = x_src if (x_src.dim() == 2 and x_src.stride(1) == 1 and x_src.stride(0) % 2 == 0
                       and x_src.data_ptr() % 8 == 0) else x_src.contiguous()
        attr = attr.contiguous()
        rot_c = rot.contiguous() if rot is not None else None
        e = src.shape[0]
        out = torch.empty(e, 4, dtype=torch.float32, device=attr.device)
        src_csr = torch.empty(e, dtype=torch.int32, device=attr.device)
        dst_csr = torch.empty(e, dtype=torch.int32, device=attr.device)
        _lib.call("hivt_edge_gather_rotate_fwd", _p(x2), x2.stride(0), _p(attr), _p(rot_c), _p(src), _p(dst), _p(perm),
                  int(rot_mod), e, _p(count), _p(out), _p(src_csr), _p(dst_csr), _stream(attr))
        ctx.save_for_backward(rot_c, src, dst, perm, count, src_csr)
        ctx.rot_mod = int(rot_mod)
        ctx.x_shape = tuple(x_src.shape)
        ctx.mark_non_differentiable(src_csr, dst_csr)
        return out, src_csr, dst_csr

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out, _g1, _g2):
        rot_c, src, dst, perm, count, src_out = ctx.saved_tensors
        e = src.shape[0]
        d_out = d_out.contiguous()
        dev = d_out.device
        d_x_edge = torch.empty(e, 4, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        d_attr = None
        if ctx.needs_input_grad[1]:
            d_attr = (torch.zeros if count is not None else torch.empty)(e, 2, dtype=torch.float32, device=dev)
        if d_x_edge is not None or d_attr is not None:
            _lib.call("hivt_edge_gather_rotate_bwd", _p(d_out), _p(rot_c), _p(src), _p(dst), _p(perm), ctx.rot_mod, e,
                      _p(count), _p(d_attr), _p(d_x_edge), _stream(d_out))
        d_x = None
        if d_x_edge is not None:
            # rows of one source summed in ascending output position: deterministic
            ns = ctx.x_shape[0]
            rowptr_s, perm_s = csr_build(src_out, ns, count)
            summed = torch.empty(ns, 4, dtype=torch.float32, device=dev)
            _lib.call("hivt_segment_sum_rows", _p(d_x_edge), 4, 4, _p(rowptr_s), _p(perm_s), ns, _p(summed), 4,
                      _stream(d_out))
            d_x = torch.zeros(ctx.x_shape, dtype=torch.float32, device=dev)
            d_x[:, :2] = summed[:, :2]
        return d_x, d_attr, None, None, None, None, None, None


def edge_gather_rotate(x_src: torch.Tensor, attr: torch.Tensor, rot: Optional[torch.Tensor], src: torch.Tensor,
                       dst: torch.Tensor, perm: Optional[torch.Tensor], rot_mod: int,
                       count: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``[x_src[src] @ R[dst % rot_mod], attr @ R[dst % rot_mod]]`` per edge in ``perm`` order -> ([E,4], src, dst).
    Rows at or beyond ``count`` (padding of fixed-capacity lists) come out as zeros / index 0."""
    return _EdgeGatherRotate.apply(x_src, attr, rot, src, dst, perm, rot_mod, count)


def edge_relpose(attr: torch.Tensor, rot: torch.Tensor, angles: torch.Tensor, src: torch.Tensor, dst: torch.Tensor,
                 perm: Optional[torch.Tensor], count: Optional[torch.Tensor] = None
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Global-interactor edge features ``[rel_pos @ R[dst], cos dtheta, sin dtheta]`` (inputs carry no grad)."""
    _need_gpu(attr, rot, angles, src, dst)
    e = src.shape[0]
    attr, rot, angles = attr.contiguous(), rot.contiguous(), angles.contiguous()
    out = torch.empty(e, 4, dtype=torch.float32, device=attr.device)
    src_csr = torch.empty(e, dtype=torch.int32, device=attr.device)
    dst_csr = torch.empty(e, dtype=torch.int32, device=attr.device)
    _lib.call("hivt_edge_relpose_fwd", _p(attr), _p(rot), _p(angles), _p(src), _p(dst), _p(perm), e, _p(count),
              _p(out), _p(src_csr), _p(dst_csr), _stream(attr))
    return out, src_csr, dst_csr


