"""Row bookkeeping of a batch: per-timestep tokens, row subsets, batch preparation, decoder inputs / outputs and the
per-code row sums behind their gradients (``rows.hip``, ``linear.hip``)."""
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
# Per-timestep tokens broadcast over agents
# --------------------------------------------------------------------------------------------
# The reference writes these as broadcasting torch.where / add / expand.  Their gradients with respect
# to the tokens are sums over the agent axis of a [T, N, D] tensor, which the framework computes with a
# multi-block reduction that clears its semaphores through a runtime memset -- a node kind that takes no
# effect in a replayed HIP graph on this stack.  Here the sums run on hivt_rowsum_by_code (fixed order).
_step_codes = {}


def _codes(steps: int, n: int, device) -> torch.Tensor:
    """int32 [steps * n]: row s * n + j carries code s."""
    key = (steps, n, device)
    if key not in _step_codes:
        if len(_step_codes) > 64:
            _step_codes.clear()
        _step_codes[key] = torch.arange(steps, dtype=torch.int32, device=device).repeat_interleave(n)
    return _step_codes[key]


def _rowsum_by_code(x2: torch.Tensor, codes: torch.Tensor, num_codes: int) -> torch.Tensor:
    m, n = x2.shape
    out = torch.empty(num_codes, n, dtype=torch.float32, device=x2.device)
    ws = _workspace(_lib.query("hivt_rowsum_by_code_workspace_bytes", m, n, num_codes), x2.device)
    _lib.call("hivt_rowsum_by_code", _p(x2), x2.stride(0), _p(codes), m, n, num_codes, _p(out), _p(ws), ws.numel(),
              _stream(x2))
    return out


def step_tokens_supported(x: torch.Tensor, steps: int) -> bool:
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[-1] <= 128 and steps <= 32 \
        and x.numel() > 0


class _MaskedStepTokens(torch.autograd.Function):
    """``where(mask[..., None], token[:, None, :], x)`` for x [T, N, D], mask [T, N], token [T, D]
    (``models/local_encoder.py:150,230``)."""

    @staticmethod
    def forward(ctx, x, mask, token):
        # a transposed mask view would hand its permuted layout to the result (and a copy to whoever reshapes it)
        mask = mask.contiguous()
        out = torch.where(mask.unsqueeze(-1), token.unsqueeze(1), x.contiguous())
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        t, n, d = dy.shape
        dy = dy.contiguous()
        dx = torch.where(mask.unsqueeze(-1), torch.zeros((), dtype=dy.dtype, device=dy.device), dy) \
            if ctx.needs_input_grad[0] else None
        dtoken = None
        if ctx.needs_input_grad[2]:
            codes = torch.where(mask.reshape(-1), _codes(t, n, dy.device), -1)
            dtoken = _rowsum_by_code(dy.view(t * n, d), codes, t)
        return dx, None, dtoken


def masked_step_tokens(x: torch.Tensor, mask: torch.Tensor, token: torch.Tensor) -> torch.Tensor:
    """Rows (t, n) with ``mask[t, n]`` replaced by ``token[t]``; x [T, N, D], mask [T, N] bool, token [T, D]."""
    return _MaskedStepTokens.apply(x, mask, token)


class _AppendTokenAddSteps(torch.autograd.Function):
    """``cat((x, last.expand(1, N, D))) + per_step[:, None, :]`` for x [T, N, D], last [D], per_step [T + 1, D]
    (the class token and the positional embedding of ``models/local_encoder.py:231-232``)."""

    @staticmethod
    def forward(ctx, x, last, per_step):
        t, n, d = x.shape
        out = torch.empty(t + 1, n, d, dtype=x.dtype, device=x.device)
        torch.add(x, per_step[:t].unsqueeze(1), out=out[:t])
        out[t] = last + per_step[t]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        s, n, d = dy.shape
        dy = dy.contiguous()
        dper = _rowsum_by_code(dy.view(s * n, d), _codes(s, n, dy.device), s)
        return dy[:s - 1], dper[s - 1].clone(), dper


def append_token_add_steps(x: torch.Tensor, last: torch.Tensor, per_step: torch.Tensor) -> torch.Tensor:
    return _AppendTokenAddSteps.apply(x, last, per_step)


# --------------------------------------------------------------------------------------------
# Row subsets: a stage that is evaluated for some rows of a [steps * agents] node set only
# --------------------------------------------------------------------------------------------
@dataclass
class RowSubset:
    """The live rows of a node set, packed to the front of ``capacity`` rows.

    ``rows`` [capacity] int32: node id of every packed row (entries at or beyond ``count`` repeat a valid id, so
    whatever is gathered through them is finite; nothing reads their results).  ``new_id`` [num_rows] int32: packed
    position of a node (nodes outside the subset: some position inside the packed range).  ``rowptr`` [capacity + 1]:
    the CSR row pointer of an edge set over the full node set, re-indexed to packed rows (rows outside the subset
    must own no edges; packed rows beyond ``count`` own none)."""
    rows: torch.Tensor
    new_id: torch.Tensor
    live: torch.Tensor           # [capacity] bool: packed row < count
    count: torch.Tensor          # 0-d int32 on the device
    capacity: int
    num_rows: int
    rowptr: Optional[torch.Tensor] = None
    overflow: Optional[torch.Tensor] = None  # 0-d bool: more live rows than capacity


def row_subset(valid: torch.Tensor, capacity: Optional[int] = None, rowptr: Optional[torch.Tensor] = None,
               count_hint: Optional[int] = None) -> RowSubset:
    """Packs the rows with ``valid`` (bool [M]) in ascending order.  Without ``capacity`` the packed size is the
    live count (``count_hint`` if the caller has already read it back, else one read-back here); with it nothing
    is read back, the packed arrays have ``capacity`` rows and ``overflow`` tells whether the live rows fit."""
    _need_gpu(valid)
    m = int(valid.shape[0])
    dev = valid.device
    pos = torch.cumsum(valid, 0, dtype=torch.int32)          # inclusive count of live rows up to each row
    count = pos[-1] if m > 0 else torch.zeros((), dtype=torch.int32, device=dev)
    if capacity is None:
        capacity = int(count_hint) if count_hint is not None else int(count.item())
    capacity = max(int(capacity), 1)
    new_id = pos - 1
    ids = torch.arange(m, dtype=torch.int32, device=dev)
    # slot `capacity + 1` swallows the rows outside the subset (and live rows that do not fit)
    slot = torch.where(valid & (new_id < capacity), new_id, capacity + 1).long()
    buf = torch.full((capacity + 2,), m, dtype=torch.int32, device=dev)
    buf.scatter_(0, slot, ids)
    packed_rowptr = rowptr.index_select(0, buf[:capacity + 1]) if rowptr is not None else None
    live = torch.arange(capacity, dtype=torch.int32, device=dev) < count
    # entries beyond the live rows repeat the first live row: gathers through them stay finite
    first = buf[0].clamp(max=max(m - 1, 0))
    rows = torch.where(live, buf[:capacity], first)
    return RowSubset(rows, new_id.clamp_(min=0, max=capacity - 1), live, count, capacity, m, packed_rowptr,
                     (count > capacity).view(()))


class _MaskedRowTokens(torch.autograd.Function):
    """``where(mask[:, None], token[code], x)`` for x [M, D], mask [M] bool, code [M] int32, token [R, D]; the
    token's gradient is a per-code row sum in a fixed order."""

    @staticmethod
    def forward(ctx, x, mask, code, token):
        out = torch.where(mask.unsqueeze(-1), token.index_select(0, code), x)
        ctx.save_for_backward(mask, code)
        ctx.codes = int(token.shape[0])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        mask, code = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.where(mask.unsqueeze(-1), torch.zeros((), dtype=dy.dtype, device=dy.device), dy) \
            if ctx.needs_input_grad[0] else None
        dtoken = _rowsum_by_code(dy, torch.where(mask, code, -1), ctx.codes) if ctx.needs_input_grad[3] else None
        return dx, None, None, dtoken


def masked_row_tokens(x: torch.Tensor, mask: torch.Tensor, code: torch.Tensor, token: torch.Tensor) -> torch.Tensor:
    """Rows with ``mask`` replaced by ``token[code]`` (x [M, D], code int32 [M] in [0, R), token [R, D], R <= 32)."""
    return _MaskedRowTokens.apply(x, mask, code, token)


class _ExpandRowsTokens(torch.autograd.Function):
    """Packed rows back on the full [T, N] node set: node (t, n) takes ``token[t]`` where ``mask[t, n]`` and packed
    row ``new_id[t * N + n]`` elsewhere (``models/local_encoder.py:230`` applied to a stage that was evaluated for
    the unmasked nodes only)."""

    @staticmethod
    def forward(ctx, x, rows, new_id, live, mask, token):
        t, n = mask.shape
        mask = mask.contiguous()
        out = torch.where(mask.unsqueeze(-1), token.unsqueeze(1), x.index_select(0, new_id).view(t, n, -1))
        ctx.save_for_backward(rows, live, mask)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        rows, live, mask = ctx.saved_tensors
        t, n, d = dy.shape
        dy = dy.contiguous().view(t * n, d)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.where(live.unsqueeze(-1), dy.index_select(0, rows),
                             torch.zeros((), dtype=dy.dtype, device=dy.device))
        dtoken = None
        if ctx.needs_input_grad[5]:
            codes = torch.where(mask.reshape(-1), _codes(t, n, dy.device), -1)
            dtoken = _rowsum_by_code(dy, codes, t)
        return dx, None, None, None, None, dtoken


def expand_rows_with_tokens(x: torch.Tensor, subset: RowSubset, mask: torch.Tensor, token: torch.Tensor) -> torch.Tensor:
    """[T, N, D] from packed rows ``x`` [capacity, D] of ``subset`` (the nodes with ``~mask``) and one token row per
    step for the masked nodes; mask [T, N] bool, token [T, D]."""
    if subset.num_rows != mask.numel() or x.shape[0] != subset.capacity:
        raise ValueError("expand_rows_with_tokens: the subset does not belong to this node set")
    return _ExpandRowsTokens.apply(x, subset.rows, subset.new_id, subset.live, mask, token)


def step_guards(nan_check: Optional[torch.Tensor], finite_check: Optional[torch.Tensor],
                extra: Optional[torch.Tensor] = None) -> torch.Tensor:
    """0-d device bool: ``nan_check`` holds a NaN, ``finite_check`` a NaN or an infinity, or ``extra`` (a
    device bool) is set -- the step's guards (``models/hivt.py:264,272,294``) in three small launches and
    without a multi-block framework reduction."""
    ref = nan_check if nan_check is not None else finite_check
    _need_gpu(ref)
    a = nan_check.contiguous() if nan_check is not None else None
    b = finite_check.contiguous() if finite_check is not None else None
    for t in (a, b):
        if t is not None and t.dtype != torch.float32:
            raise TypeError("step_guards checks float32 tensors")
    ex = extra.view(torch.uint8) if (extra is not None and extra.dtype == torch.bool) else extra
    flag = torch.empty(1, dtype=torch.int32, device=ref.device)
    _lib.call("hivt_step_guards", _p(a), a.numel() if a is not None else 0, _p(b), b.numel() if b is not None else 0,
              _p(ex), _p(flag), _stream(ref))
    return flag.view(torch.bool)[0]


# --------------------------------------------------------------------------------------------
# Row bookkeeping of a batch in single launches (hivt_gan_amd/csrc/rows.hip)
# --------------------------------------------------------------------------------------------
def _u8(t: torch.Tensor) -> torch.Tensor:
    """A bool / uint8 tensor as uint8 whose last dimension is contiguous (a column slice of a 2-D mask stays a view:
    the kernels take the row stride)."""
    if not (t.dim() == 2 and t.stride(1) == 1):
        t = t.contiguous()
    return t.view(torch.uint8) if t.dtype == torch.bool else t


_unit_scalars = {}


def unit_scalar(ref: torch.Tensor) -> torch.Tensor:
    """A cached 0-d float32 one on ``ref``'s device: the seed gradient of a backward pass (``loss.backward(unit)``)
    without the fill that ``backward()`` issues for it on every call."""
    key = str(ref.device)
    t = _unit_scalars.get(key)
    if t is None:
        t = _unit_scalars[key] = torch.ones((), dtype=torch.float32, device=ref.device)
    return t


def batch_prepare_supported(data, historical_steps: int) -> bool:
    """Whether ``batch_prepare`` applies: a GPU batch with the reference's tensor types."""
    x = data.get("x")
    if x is None or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 3 or x.shape[2] != 2:
        return False
    pm = data.get("padding_mask")
    if pm is None or pm.dtype not in (torch.bool, torch.uint8) or pm.shape[1] < historical_steps:
        return False
    for key in ("is_intersections", "turn_directions", "traffic_controls"):
        t = data.get(key)
        if t is None or t.dtype not in (torch.uint8, torch.bool):
            return False
    return data.get("rotate_angles") is not None and x.shape[1] >= historical_steps


def batch_prepare(data, historical_steps: int) -> dict:
    """Everything the step derives from the raw batch before the first layer, in one launch: ``rotate_mat``
    [N, 2, 2], the targets rotated into the agents' frames (``y``, or None), ``x_rot`` [T * N, 2] (step-major rotated
    displacements; ``x_steps``: the un-rotated ones in the same order), ``lane_code`` int32 [L] and ``keep_w`` [N] (1
    where an agent has a valid future step)."""
    x = data["x"]
    n = int(x.shape[0])
    T = int(historical_steps)
    dev = x.device
    pm = _u8(data["padding_mask"])
    hf = int(pm.shape[1]) - T
    y = data.get("y")
    if y is not None:
        if y.dtype != torch.float32 or tuple(y.shape) != (n, hf, 2):
            raise ValueError(f"y must be float32 [N, {hf}, 2], got {y.dtype} {tuple(y.shape)}")
        y = y.contiguous()
    ang = data["rotate_angles"].contiguous()
    if ang.dtype != torch.float32 or ang.shape != (n,):
        raise ValueError("rotate_angles must be float32 [N]")
    if x.stride(2) != 1:
        x = x.contiguous()
    lanes = [_u8(data[k]) for k in ("is_intersections", "turn_directions", "traffic_controls")]
    num_lanes = int(lanes[0].shape[0])
    rot = torch.empty(n, 2, 2, dtype=torch.float32, device=dev)
    y_rot = torch.empty(n, hf, 2, dtype=torch.float32, device=dev) if y is not None else None
    x_rot = torch.empty(T * n, 2, dtype=torch.float32, device=dev)
    x_steps = torch.empty(T * n, 2, dtype=torch.float32, device=dev)
    lane_code = torch.empty(num_lanes, dtype=torch.int32, device=dev)
    keep_w = torch.empty(n, dtype=torch.float32, device=dev)
    _lib.call("hivt_batch_prepare", _p(ang), _p(y), _p(x), x.stride(0), x.stride(1), _p(pm), pm.stride(0), _p(lanes[0]),
              _p(lanes[1]), _p(lanes[2]), n, T, hf, num_lanes, _p(rot), _p(y_rot), _p(x_rot), _p(x_steps), _p(lane_code),
              _p(keep_w), _stream(x))
    return {"rotate_mat": rot, "y": y_rot, "x_rot": x_rot, "x_steps": x_steps, "lane_code": lane_code, "keep_w": keep_w}


def row_subset_fixed(padding_mask: torch.Tensor, steps: int, capacity: int, rowptr: Optional[torch.Tensor],
                     edge_counts=(), edge_caps=()) -> RowSubset:
    """``row_subset`` of the (step, agent) nodes with ``~padding_mask[:, :steps]`` at a fixed capacity, in two
    launches and without a read-back; ``overflow`` also covers the given device edge counters against their
    capacities (the flag a fixed-capacity step discards its batch by)."""
    _need_gpu(padding_mask)
    pm = _u8(padding_mask)
    n = int(pm.shape[0])
    dev = pm.device
    capacity = max(int(capacity), 1)
    m = n * int(steps)
    rows = torch.empty(capacity, dtype=torch.int32, device=dev)
    new_id = torch.empty(m, dtype=torch.int32, device=dev)
    live = torch.empty(capacity, dtype=torch.bool, device=dev)
    count = torch.empty(2, dtype=torch.int32, device=dev)
    overflow = torch.empty(1, dtype=torch.bool, device=dev)
    packed = torch.empty(capacity + 1, dtype=torch.int32, device=dev) if rowptr is not None else None
    if rowptr is not None and (rowptr.dtype != torch.int32 or rowptr.shape[0] != m + 1):
        raise ValueError("rowptr must be int32 with steps * N + 1 entries")
    cnt = list(edge_counts) + [None] * (3 - len(edge_counts))
    cap = [int(c) for c in edge_caps] + [0] * (3 - len(edge_caps))
    ws = _workspace(_lib.query("hivt_row_subset_workspace_bytes", n, int(steps)), dev)
    _lib.call("hivt_row_subset", _p(pm), pm.stride(0), n, int(steps), capacity, _p(rowptr), _p(cnt[0]), cap[0], _p(cnt[1]),
              cap[1], _p(cnt[2]), cap[2], _p(rows), _p(new_id), _p(live), _p(count), _p(packed), _p(overflow), _p(ws),
              ws.numel(), _stream(pm))
    return RowSubset(rows, new_id, live, count[0], capacity, m, packed, overflow.view(()))


def center_rows(x_rot: torch.Tensor, bos_mask: torch.Tensor, rows: torch.Tensor, n: int, steps: int
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per packed (step, agent) node: its rotated displacement [capacity, 2] and its begin-of-sequence code (the
    step where ``bos_mask[agent, step]``, else -1)."""
    bos = _u8(bos_mask)
    cap = int(rows.shape[0])
    xc = torch.empty(cap, 2, dtype=torch.float32, device=x_rot.device)
    code = torch.empty(cap, dtype=torch.int32, device=x_rot.device)
    _lib.call("hivt_center_rows", _p(x_rot), _p(bos), bos.stride(0), _p(rows), int(n), int(steps), cap, _p(xc), _p(code),
              _stream(x_rot))
    return xc, code


class _RowsPutTokens(torch.autograd.Function):
    """``where(code >= 0, token[code], x)`` for x [M, D], code int32 [M] (-1: keep the row), token [R <= 32, D]."""

    @staticmethod
    def forward(ctx, x, code, token):
        _need_gpu(x, code, token)
        x2 = _rows(x)
        m, d = x2.shape
        token_c = token.contiguous()
        out = torch.empty(m, d, dtype=torch.float32, device=x.device)
        _lib.call("hivt_rows_put_tokens", _p(x2), x2.stride(0), _p(code), _p(token_c), m, d, 0, _p(out), _stream(x))
        ctx.save_for_backward(code)
        ctx.codes = int(token.shape[0])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (code,) = ctx.saved_tensors
        dy = _rows(dy)
        m, d = dy.shape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(m, d, dtype=torch.float32, device=dy.device)
            _lib.call("hivt_rows_put_tokens", _p(dy), dy.stride(0), _p(code), None, m, d, 0, _p(dx), _stream(dy))
        dtoken = _rowsum_by_code(dy, code, ctx.codes) if ctx.needs_input_grad[2] else None
        return dx, None, dtoken


def rows_put_tokens(x: torch.Tensor, code: torch.Tensor, token: torch.Tensor) -> torch.Tensor:
    """Rows whose ``code`` is non-negative replaced by ``token[code]`` (one launch; the token's gradient is a
    per-code row sum in a fixed order)."""
    if x.shape[-1] % 4 != 0 or token.shape[0] > 32:
        raise _unsupported(f"token rows of width {x.shape[-1]} over {token.shape[0]} codes")
    return _RowsPutTokens.apply(x, code, token)


class _TemporalInput(torch.autograd.Function):
    """Input of the temporal encoder, ``models/local_encoder.py:230-232``: for step s < T row (s, n) is the padding
    token of step s where ``padding_mask[n, s]``, else the agent-agent stage's row (packed rows looked up through
    ``new_id``), row (T, n) the class token; the positional row of the step is added to all of them."""

    @staticmethod
    def forward(ctx, x, new_id, rows, count, padding_mask, pad_token, cls_token, pos_embed):
        _need_gpu(x, padding_mask, pad_token)
        pm = _u8(padding_mask)
        n = int(pm.shape[0])
        steps, d = int(pad_token.shape[0]), int(pad_token.shape[1])
        x2 = x.reshape(-1, d).contiguous()
        pad_c, cls_c, pos_c = pad_token.contiguous(), cls_token.contiguous(), pos_embed.contiguous()
        out = torch.empty(steps + 1, n, d, dtype=torch.float32, device=x.device)
        pad_code = torch.empty((steps + 1) * n, dtype=torch.int32, device=x.device)
        _lib.call("hivt_temporal_input_fwd", _p(x2), _p(new_id), _p(pm), pm.stride(0), _p(pad_c), _p(cls_c), _p(pos_c),
                  steps, n, d, _p(out), _p(pad_code), _stream(x))
        ctx.save_for_backward(rows, count, pad_code)
        ctx.cfg = (steps, n, d, tuple(x.shape))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        rows, count, pad_code = ctx.saved_tensors
        steps, n, d, x_shape = ctx.cfg
        dy2 = dy.contiguous().view((steps + 1) * n, d)
        dev = dy2.device
        dx = None
        if ctx.needs_input_grad[0]:
            if rows is not None:
                cap = int(rows.shape[0])
                dx = torch.empty(cap, d, dtype=torch.float32, device=dev)
                _lib.call("hivt_gather_live_rows", _p(dy2), _p(rows), _p(count), cap, d, _p(dx), _stream(dy2))
            else:  # every (step, agent) row was an input row: the gradient is dy outside the padded rows
                dx = torch.empty(steps * n, d, dtype=torch.float32, device=dev)
                _lib.call("hivt_rows_put_tokens", _p(dy2), d, _p(pad_code), None, steps * n, d, 0, _p(dx), _stream(dy2))
            dx = dx.view(x_shape)
        dpad = _rowsum_by_code(dy2, pad_code, steps) if ctx.needs_input_grad[5] else None
        dpos = dcls = None
        if ctx.needs_input_grad[6] or ctx.needs_input_grad[7]:
            dpos = _rowsum_by_code(dy2, _codes(steps + 1, n, dev), steps + 1)
            dcls = dpos[steps] if ctx.needs_input_grad[6] else None
        return dx, None, None, None, None, dpad, dcls, dpos


def temporal_input(x: torch.Tensor, subset: Optional[RowSubset], padding_mask: torch.Tensor, pad_token: torch.Tensor,
                   cls_token: torch.Tensor, pos_embed: torch.Tensor) -> torch.Tensor:
    """[T + 1, N, D] input of the temporal encoder in one launch.  ``x``: the agent-agent stage's rows, packed
    ([subset.capacity, D]) with ``subset`` or all of them ([T, N, D]) without; ``padding_mask`` [N, >= T]; pad_token
    [T, D], cls_token [D], pos_embed [T + 1, D]."""
    if subset is not None:
        if subset.num_rows != padding_mask.shape[0] * pad_token.shape[0] or x.shape[0] != subset.capacity:
            raise ValueError("temporal_input: the subset does not belong to this node set")
        return _TemporalInput.apply(x, subset.new_id, subset.rows, subset.count.view(1), padding_mask, pad_token,
                                    cls_token, pos_embed)
    return _TemporalInput.apply(x, None, None, None, padding_mask, pad_token, cls_token, pos_embed)


class _CategoryTable(torch.autograd.Function):

    @staticmethod
    def forward(ctx, a, b, c):
        _need_gpu(a, b, c)
        a, b, c = a.contiguous(), b.contiguous(), c.contiguous()
        d = int(a.shape[1])
        out = torch.empty(a.shape[0] * b.shape[0] * c.shape[0], d, dtype=torch.float32, device=a.device)
        _lib.call("hivt_category_table_fwd", _p(a), _p(b), _p(c), a.shape[0], b.shape[0], c.shape[0], d, _p(out),
                  _stream(a))
        ctx.sizes = (int(a.shape[0]), int(b.shape[0]), int(c.shape[0]), d)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        na, nb, nc, d = ctx.sizes
        g = g.contiguous()
        da = torch.empty(na, d, dtype=torch.float32, device=g.device)
        db = torch.empty(nb, d, dtype=torch.float32, device=g.device)
        dc = torch.empty(nc, d, dtype=torch.float32, device=g.device)
        _lib.call("hivt_category_table_bwd", _p(g), na, nb, nc, d, _p(da), _p(db), _p(dc), _stream(g))
        return da, db, dc


def category_table(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """``[(a[i] + b[j]) + c[k]]`` for all (i, j, k), row ``(i * len(b) + j) * len(c) + k``: the summed categorical
    embeddings of ``models/local_encoder.py:361-363`` as one small table (one launch, one for its gradient)."""
    return _CategoryTable.apply(a, b, c)


class _DecoderOutput(torch.autograd.Function):

    @staticmethod
    def forward(ctx, loc, raw, modes, agent_major, min_scale):
        _need_gpu(loc, raw)
        loc2, raw2 = _rows2(loc), _rows2(raw)
        m, two_h = loc2.shape
        h = two_h // 2
        out = torch.empty(modes, m // modes, h, 4, dtype=torch.float32, device=loc.device)
        f = modes if agent_major else 1
        _lib.call("hivt_decoder_output_fwd", _p(loc2), loc2.stride(0), _p(raw2), raw2.stride(0), m, h, f, float(min_scale),
                  _p(out), _stream(loc))
        ctx.save_for_backward(raw2)
        ctx.cfg = (m, h, f, float(min_scale), tuple(loc.shape), tuple(raw.shape))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (raw2,) = ctx.saved_tensors
        m, h, f, min_scale, loc_shape, raw_shape = ctx.cfg
        g = g.contiguous()
        d_loc = torch.empty(m, 2 * h, dtype=torch.float32, device=g.device)
        d_raw = torch.empty(m, 2 * h, dtype=torch.float32, device=g.device)
        _lib.call("hivt_decoder_output_bwd", _p(g), _p(raw2), raw2.stride(0), m, h, f, min_scale, _p(d_loc), _p(d_raw),
                  _stream(g))
        return d_loc.view(loc_shape), d_raw.view(raw_shape), None, None, None


def decoder_output(loc: torch.Tensor, scale_raw: torch.Tensor, modes: int, agent_major: bool,
                   min_scale: float = 0.05) -> torch.Tensor:
    """``cat(loc, clamp(elu(scale_raw) + 1, min=min_scale))`` viewed [modes, N, H, 4] (``models/decoder.py:131-139``)
    from the two heads' outputs [modes * N, 2 H]; ``agent_major``: their rows are ordered (agent, mode)."""
    return _DecoderOutput.apply(loc, scale_raw, int(modes), bool(agent_major), float(min_scale))


