"""torch.autograd.Functions and tensor-level wrappers over the C ABI of ``libhivt_mi355.so``.

Everything here requires tensors resident on the GPU; there is no CPU implementation behind these
functions.  PyTorch supplies device memory, the current stream and autograd bookkeeping only.

This module holds the attention, dense-layer and block-level nodes together with the state they share (the queue of
deferred weight gradients, the dropout seeds, the timing hook).  The stateless wrappers live next to it and are
re-exported here, so that ``ops.<name>`` reaches everything: ``ops_edges`` (edge construction, gather + rotate),
``ops_rows`` (tokens, row subsets, batch preparation, decoder inputs / outputs), ``ops_losses`` (supervised loss,
adversarial objectives), ``ops_spectral`` (spectral normalisation of the critics' layers).
"""
from __future__ import annotations

import contextlib
import ctypes
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from ._opsbase import (UnsupportedShapeError, _ld, _need_gpu, _p, _ptr_array, _rows, _rows2, _stream, _unsupported,  # noqa: F401
                       _workspace)
from .ops_edges import (  # noqa: F401  (every name stays reachable as ops.<name>)
    edges_temporal, edges_radius, csr_build, edges_all_pairs_csr, EdgeSet, make_edge_set,
    _EdgeGatherRotate, edge_gather_rotate, edge_relpose)
from .ops_losses import (  # noqa: F401  (every name stays reachable as ops.<name>)
    _SupervisedLoss, supervised_loss, supervised_loss_packed, _HingeR1Loss, hinge_r1_loss,
    _WeightedMean, weighted_mean, _HingeR1Multi, hinge_r1_multi, _GeneratorTotal, generator_total)
from .ops_spectral import (  # noqa: F401  (every name stays reachable as ops.<name>)
    _round4, _aligned_parts, _SpectralNormalize, _SpectralNormalizeRounds,
    spectral_normalize_rounds, spectral_normalize)
from .ops_rows import (  # noqa: F401  (every name stays reachable as ops.<name>)
    _step_codes, _codes, _rowsum_by_code, step_tokens_supported, _MaskedStepTokens,
    masked_step_tokens, _AppendTokenAddSteps, append_token_add_steps, RowSubset, row_subset,
    _MaskedRowTokens, masked_row_tokens, _ExpandRowsTokens, expand_rows_with_tokens, step_guards,
    _u8, _unit_scalars, unit_scalar, batch_prepare_supported, batch_prepare, row_subset_fixed,
    center_rows, _RowsPutTokens, rows_put_tokens, _TemporalInput, temporal_input, _CategoryTable,
    category_table, _DecoderOutput, decoder_output)



# --------------------------------------------------------------------------------------------
# Kernels (b)+(c): segment softmax attention
# --------------------------------------------------------------------------------------------
# When set to a list, every segment-attention launch appends
# (kind, tag, rows, rowptr, D, H, has_node_terms, start_event, end_event) -- events carrying the
# kernel's own start / stop time stamps, and the launch's rowptr (device tensor: fixed-capacity lists
# hold fewer live edges than rows of K|V, and rows without edges move fewer bytes) -- so a benchmark can
# price the kernel inside a real step.
SEG_ATTN_TIMING = None


def _timed_call(kind, tag, n, live, d_model, heads, node_terms, dev, name, *args, grad_bytes=4):
    """``grad_bytes``: bytes per element of the dk / dv rows a backward launch writes (2 for bfloat16 rows)."""
    rec = SEG_ATTN_TIMING
    if rec is None:
        _lib.call(name, *args)
        return
    stream = torch.cuda.current_stream(dev)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record(stream)  # creates the underlying events; the timed launch re-records them with the
    ev1.record(stream)  # kernel's own start / stop time stamps
    _lib.call(name + "_timed", *args, int(ev0.cuda_event), int(ev1.cuda_event))
    rec.append((kind, tag, n, live, d_model, heads, node_terms, ev0, ev1, grad_bytes))


# Dropout seeds.  Every call draws a host-side seed (CPU generator: no device sync, reproducible
# under torch.manual_seed).  When DEVICE_SEED is set (a uint64-as-int64 device scalar owned by the
# caller) the kernels add its current value to a per-call constant instead, so a caller that
# re-issues identical launches still draws new masks as long as it advances the scalar.
#
# Inside ``seed_scope()`` (one training step) all seeds derive from ONE draw of the CPU generator and a
# call counter, which saves the forty-odd generator calls of a step; seeding torch before the scope
# still reproduces every mask of the scope.
DEVICE_SEED: Optional[torch.Tensor] = None
_seed_calls = 0
_scope_base: Optional[int] = None
_scope_calls = 0
_SEED_MASK = (1 << 62) - 1
_GOLDEN = 0x9E3779B97F4A7C15


@contextlib.contextmanager
def seed_scope():
    global _scope_base, _scope_calls
    outer = (_scope_base, _scope_calls)
    _scope_base = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    _scope_calls = 0
    try:
        yield
    finally:
        _scope_base, _scope_calls = outer


def _fresh_seed() -> int:
    global _seed_calls, _scope_calls
    if DEVICE_SEED is not None:
        _seed_calls += 1
        return (_seed_calls * _GOLDEN) & _SEED_MASK
    if _scope_base is not None:
        _scope_calls += 1
        return (_scope_base + _scope_calls * _GOLDEN) & _SEED_MASK
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def _seed_ptr(dropout_p: float):
    return _p(DEVICE_SEED) if (DEVICE_SEED is not None and dropout_p > 0.0) else None


class _SegAttn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, q, kv, kv_node, rowptr, src, rowptr_by_src, perm_by_src, num_heads, dropout_p, seed, tag,
                padded, interleaved=False, row_order=None):
        _need_gpu(q, kv, rowptr)
        n, d_model = q.shape
        e = kv.shape[0]
        if kv.shape[1] != 2 * d_model:
            raise ValueError(f"kv must be [E, 2*D] (K | V), got {tuple(kv.shape)} for D={d_model}")
        if rowptr.shape[0] != n + 1:
            raise ValueError("rowptr must have num_targets + 1 entries")
        if row_order is not None and (row_order.dtype != torch.int32 or row_order.shape != (n,)
                                      or not row_order.is_contiguous()):
            raise ValueError("row_order must be a contiguous int32 permutation of the target rows")
        q = _rows(q)  # (may be a column block of a wider product: its row stride goes to the kernel)
        kv = _rows(kv)
        if kv_node is not None:
            kv_node = _rows(kv_node)
            if src is None or src.shape[0] != e:
                raise ValueError("node terms need the per-edge source ids in CSR order")
        out = torch.empty(n, d_model, dtype=torch.float32, device=q.device)
        m = torch.empty(n, num_heads, dtype=torch.float32, device=q.device)
        den = torch.empty(n, num_heads, dtype=torch.float32, device=q.device)
        fb = 4 * d_model
        _timed_call("fwd", tag, n, rowptr, d_model, num_heads, kv_node is not None, q.device,
                    "hivt_seg_attn_fwd", _p(q), _ld(q), _p(kv), _p(kv, fb), kv.stride(0) if e > 0 else 2 * d_model,
                  _p(kv_node), _p(kv_node, fb) if kv_node is not None else None,
                  kv_node.stride(0) if kv_node is not None else 0, _p(src) if kv_node is not None else None,
                    _p(rowptr), _p(row_order), n, e, d_model, num_heads, float(dropout_p), int(seed),
                    _seed_ptr(dropout_p), _p(out),
                    _p(m), _p(den), ROWS_INTERLEAVED if interleaved else 0, _stream(q))
        ctx.save_for_backward(q, kv, kv_node, rowptr, src, rowptr_by_src, perm_by_src, out, m, den,
                              DEVICE_SEED if dropout_p > 0.0 else None, row_order)
        ctx.cfg = (num_heads, float(dropout_p), int(seed), tag, bool(padded), bool(interleaved))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out):
        q, kv, kv_node, rowptr, src, rowptr_by_src, perm_by_src, out, m, den, seed_dev, row_order = ctx.saved_tensors
        num_heads, dropout_p, seed, tag, padded, interleaved = ctx.cfg
        n, d_model = q.shape
        e = kv.shape[0]
        d_out = d_out.contiguous()
        # (set by a caller that runs this node's two halves itself, see _EdgeAttention / _AttentionBlock: where the
        # query gradient goes, and whether that caller consumes the key | value gradient as bfloat16)
        dq = getattr(ctx, "dq_out", None)
        if dq is None:
            dq = torch.empty(n, d_model, dtype=torch.float32, device=q.device)
        half = bool(getattr(ctx, "dkv_half", False))
        if half and (kv_node is not None or interleaved):
            raise RuntimeError("bfloat16 key | value gradients: plain rows only")
        dkv = torch.empty(e, 2 * d_model, dtype=torch.bfloat16 if half else torch.float32, device=q.device)
        fb_kv = 4 * d_model
        fb = 2 * d_model if half else 4 * d_model
        layout = (ROWS_INTERLEAVED if interleaved else 0) | (DKV_BF16 if half else 0)
        _timed_call("bwd", tag, n, rowptr, d_model, num_heads, kv_node is not None, q.device,
                    "hivt_seg_attn_bwd", _p(d_out), _p(q), _ld(q), _p(kv), _p(kv, fb_kv),
                    kv.stride(0) if e > 0 else 2 * d_model,
                    _p(kv_node), _p(kv_node, fb_kv) if kv_node is not None else None,
                    kv_node.stride(0) if kv_node is not None else 0, _p(src) if kv_node is not None else None,
                    _p(rowptr), _p(row_order), _p(out), _p(m), _p(den), n, e, d_model, num_heads, dropout_p, seed,
                    _p(seed_dev),
                    _p(dq), _ld(dq), _p(dkv), _p(dkv, fb), 2 * d_model, 1 if padded else 0, layout, _stream(q),
                    grad_bytes=2 if half else 4)
        dkv_node = None
        if kv_node is not None and ctx.needs_input_grad[2]:
            ns = kv_node.shape[0]
            dkv_node = torch.empty(ns, 2 * d_model, dtype=torch.float32, device=q.device)
            _lib.call("hivt_segment_sum_rows", _p(dkv), 2 * d_model, 2 * d_model, _p(rowptr_by_src), _p(perm_by_src),
                      ns, _p(dkv_node), 2 * d_model, _stream(q))
        return dq, dkv, dkv_node, None, None, None, None, None, None, None, None, None, None, None


ROWS_INTERLEAVED = 1 << 16  # layout flag in the dtype argument of hivt_seg_attn_fwd / _bwd
DKV_BF16 = 1 << 17          # hivt_seg_attn_bwd: dk / dv are bfloat16 rows


def seg_attn_supported(d_model: int, num_heads: int) -> bool:
    """Widths the segment-attention kernels are built for."""
    if d_model not in (16, 32, 64, 128, 256) or num_heads <= 0 or d_model % (4 * num_heads) != 0:
        return False
    return d_model // num_heads in (4, 8, 16, 32)


def seg_attn(q: torch.Tensor, kv: torch.Tensor, rowptr: torch.Tensor, num_heads: int, dropout_p: float = 0.0,
             training: bool = False, kv_node: Optional[torch.Tensor] = None, src: Optional[torch.Tensor] = None,
             rowptr_by_src: Optional[torch.Tensor] = None, perm_by_src: Optional[torch.Tensor] = None,
             tag: str = "", padded: bool = False, interleaved: bool = False,
             row_order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Segment-softmax attention over CSR rows.

    ``q`` [N,D]; ``kv`` [E,2D] holds K | V per edge in CSR order; optional ``kv_node`` [Ns,2D] adds the
    source node's K | V (``src`` gives the source id of every CSR-ordered edge, and the CSR over
    sources is needed for its gradient).  ``padded`` says that ``kv`` has more rows than ``rowptr``
    covers (fixed-capacity edge lists): the gradient rows of that padding are then zeroed.
    ``interleaved``: every target has E / N edges and the j-th edge of target n is row ``j * N + n`` of ``kv``
    (a [steps, N, 2D] tensor whose targets attend over their own steps); no node terms, no padding.
    Returns the aggregated messages [N,D]."""
    p = float(dropout_p) if training else 0.0
    seed = _fresh_seed() if p > 0.0 else 0
    if kv_node is not None and kv_node.requires_grad and (rowptr_by_src is None or perm_by_src is None):
        raise ValueError("gradient of the node terms needs the CSR over sources")
    if interleaved and (kv_node is not None or padded or q.shape[0] == 0 or kv.shape[0] % q.shape[0] != 0):
        raise ValueError("interleaved rows: E must be a multiple of N, without node terms or padding")
    return _SegAttn.apply(q, kv, kv_node, rowptr, src, rowptr_by_src, perm_by_src, int(num_heads), p, seed, tag,
                          padded, bool(interleaved), row_order)


# --------------------------------------------------------------------------------------------
# Gated update
# --------------------------------------------------------------------------------------------
class _GatedUpdate(torch.autograd.Function):

    @staticmethod
    def forward(ctx, agg, ih, hh_self):
        _need_gpu(agg, ih, hh_self)
        n, d_model = agg.shape
        agg, ih, hs = _rows(agg), _rows(ih), _rows(hh_self)
        if hs.shape != (n, 2 * d_model) or ih.shape != (n, d_model):
            raise ValueError("gated_update expects agg [N,D], ih [N,D], hh_self [N,2D]")
        out = torch.empty(n, d_model, dtype=torch.float32, device=agg.device)
        _lib.call("hivt_gated_update_fwd", _p(agg), agg.stride(0), _p(ih), ih.stride(0), _p(hs), hs.stride(0),
                  _p(hs, 4 * d_model), hs.stride(0), n, d_model, _p(out), _stream(agg))
        ctx.save_for_backward(agg, ih, hs)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out):
        agg, ih, hs = ctx.saved_tensors
        n, d_model = agg.shape
        d_out = d_out.contiguous()
        d_agg = torch.empty(n, d_model, dtype=torch.float32, device=agg.device)
        d_ih = torch.empty(n, d_model, dtype=torch.float32, device=agg.device)
        d_hs = torch.empty(n, 2 * d_model, dtype=torch.float32, device=agg.device)
        _lib.call("hivt_gated_update_bwd", _p(d_out), _p(agg), agg.stride(0), _p(ih), ih.stride(0), _p(hs),
                  hs.stride(0), _p(hs, 4 * d_model), hs.stride(0), n, d_model, _p(d_agg), _p(d_ih), _p(d_hs),
                  2 * d_model, _p(d_hs, 4 * d_model), 2 * d_model, _stream(agg))
        return d_agg, d_ih, d_hs


def gated_update(agg: torch.Tensor, ih: torch.Tensor, hh_self: torch.Tensor) -> torch.Tensor:
    """``agg + sigmoid(ih + hh) * (self - agg)`` with ``hh_self = [hh | self]`` of shape [N, 2D]."""
    return _GatedUpdate.apply(agg, ih, hh_self)


# --------------------------------------------------------------------------------------------
# Dense projections on the fp32 matrix cores
# --------------------------------------------------------------------------------------------
EPI_NONE, EPI_RELU, EPI_LN, EPI_LN_RELU = 0, 1, 2, 3


def linear_fused_supported(n_out: int, layer_norm: bool) -> bool:
    """Whether hivt_linear_fwd has a kernel for this output width / epilogue."""
    return (not layer_norm) or n_out <= 128


def _pass_through_grad(dy: torch.Tensor, shape, to_leaf: bool) -> torch.Tensor:
    """Gradient of an operand that was only added to the output: the incoming gradient itself.  A leaf
    keeps what it is handed as its ``.grad`` and later accumulates into it in place, so it gets a copy
    (a view sharing ``dy``'s memory would let that accumulation reach every other holder of ``dy``)."""
    g = dy.reshape(shape)
    return g.clone() if to_leaf else g


_matmul_mask = 0  # what set_matmul_precision last set


def _half_gradients() -> bool:
    """True while both backward products round their operands to bfloat16: a gradient that only those products
    read may then be kept in memory as bfloat16 (the rounding happens where it is written instead of where it is
    read; the products see the same values)."""
    return (_matmul_mask & 6) == 6


HALF_DY, HALF_DX, MASK_BITS = 1, 2, 4
EPI_SIGN_BITS = 0x100

_PRECISIONS = {"fp32": 0, "bf16": 14, "bf16-plain": 7, "bf16-fwd": 1, "bf16-bwd": 6, "bf16-split-fwd": 8}


def set_matmul_precision(precision: str) -> str:
    """Operand precision of the dense products, process-wide; returns the previous setting.

    ``"fp32"`` (default): exact fp32 products.  ``"bf16"``: operands of every dense product (forward,
    data gradient, weight gradient) rounded to bfloat16, fp32 accumulation.  ``"bf16-fwd-dgrad"`` /
    ``"bf16-fwd"``: only those products take bf16 operands, the others stay exact."""
    global _matmul_mask
    if precision not in _PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
    prev = _lib.value("hivt_set_matmul_bf16", _PRECISIONS[precision])
    _matmul_mask = _PRECISIONS[precision]
    for name, mask in _PRECISIONS.items():
        if mask == prev:
            return name
    return "bf16"


def linear_wgrad_group(terms) -> None:
    """Several weight gradients in one library call.  ``terms``: tuples ``(dy [M, N], x [M, K], dw [N, K],
    dbias [N] or None, accumulate, joins)`` -- ``dw (+)= dy^T x``, ``dbias (+)= column sums of dy``; ``joins``:
    the term adds to the same ``dw`` as the term before it.  Results equal one ``hivt_linear_wgrad`` call per
    term bit for bit; terms of a few thousand rows share launches."""
    if not terms:
        return
    flat = []
    for dy, x, dw, db, accumulate, joins in terms:
        m, n = dy.shape
        k = x.shape[1]
        if x.shape[0] != m or tuple(dw.shape) != (n, k) or not dw.is_contiguous() or (db is not None and db.numel() != n):
            raise ValueError("linear_wgrad_group: operand shapes do not fit")
        flat += [dy.data_ptr(), dy.stride(0) if m > 1 else n, x.data_ptr(), x.stride(0) if m > 1 else k, m, n, k,
                 dw.data_ptr(), db.data_ptr() if db is not None else 0, (1 if accumulate else 0) | (2 if joins else 0)]
    _launch_wgrad_group(flat, len(terms), terms[0][0].device)


def _launch_wgrad_group(flat, count: int, dev) -> None:
    table = (ctypes.c_int64 * len(flat))(*flat)
    ws = _workspace(_lib.value("hivt_linear_wgrad_group_workspace_bytes", table, count), dev)
    _lib.call("hivt_linear_wgrad_group", table, count, _p(ws), ws.numel(), torch._C._cuda_getCurrentRawStream(dev.index))


# Weight gradients are leaves of the backward pass: nothing but the optimizer reads them.  Inside
# ``deferred_weight_grads()`` their products (on their own a launch each for the product and for the sum over
# its chunks: all latency for layers of a few thousand rows, ramp-up and partial-sum traffic for long ones) are
# collected while the backward pass runs and issued together when it ends (a callback of the autograd engine),
# a few launches for all of them.  The operands stay alive until then.
#
# Contract.  The gradient tensors handed to autograd exist from the start and are FILLED by that call, so autograd
# must do nothing with them but adopt them as the parameters' ``.grad``.  A product is therefore deferred only if,
# when its node runs, every parameter of the node is a leaf whose ``.grad`` is None and which carries no hooks
# (a populated ``.grad`` -- gradient accumulation, ``zero_grad(set_to_none=False)`` -- would have the unfilled
# buffer added to it; such products run on the spot).  What cannot be seen from inside a node -- one parameter
# feeding two nodes of a graph, ``retain_graph`` replays, ``create_graph=True`` -- is checked when the queue is
# issued: every buffer must by then BE the ``.grad`` of one of its node's parameters, otherwise the pass raises
# instead of training on a sum that contains an unfilled buffer.  Nothing may read a parameter's ``.grad`` from a
# hook before the pass has ended -- except a hook registered through ``trust_grad_hook``, which issues the queue
# (``flush_weight_grads(final=False)``) before it reads (the data-parallel bucket's chunk-complete hook).
_deferred: Optional[list] = None


class deferred_weight_grads:
    """Context manager; see above.  Not re-entrant across threads: one training step at a time."""

    def __enter__(self):
        global _deferred, _deferred_rows
        self._prev = (_deferred, _deferred_rows)
        _deferred, _deferred_rows = [], []
        return self

    def __exit__(self, *exc):
        global _deferred, _deferred_rows
        try:
            if exc[0] is None:
                flush_weight_grads()
        finally:
            del _pending_checks[:]
            _deferred, _deferred_rows = self._prev
        return False


def _check_adopted(entries) -> None:
    """Every deferred gradient buffer must have become the ``.grad`` of one of its node's parameters."""
    for addresses, params in entries:
        grads = {p.grad.data_ptr() for p in params if p.grad is not None}
        # (views of one buffer -- the two halves of a pair of layers -- are adopted as views: same addresses)
        for a in addresses:
            if a and a not in grads:
                raise RuntimeError(
                    "deferred_weight_grads(): a weight gradient that was queued for the end of the backward pass was "
                    "not adopted as its parameter's .grad (a parameter used by two nodes of one graph, a tensor hook, "
                    "retain_grad or create_graph): its buffer was read before it was filled.  Run this backward pass "
                    "outside deferred_weight_grads() (GanTrainer.defer_weight_grads = False)")


_pending_checks: list = []   # adoption checks of products issued before the end of the pass (see flush_weight_grads)
_trusted_hooks: set = set()  # grad-ready hooks that read no gradient before having flushed the queue


def trust_grad_hook(hook) -> None:
    """Declares a post-accumulate-grad hook safe for deferred weight gradients: it reads no ``.grad`` unless it has
    called ``flush_weight_grads(final=False)`` first (the data-parallel bucket's chunk-complete hook does)."""
    _trusted_hooks.add(hook)


def flush_weight_grads(final: bool = True) -> None:
    """Issues the collected weight gradients (runs by itself at the end of every backward pass).  ``final=False``:
    called from inside the pass (a chunk of parameters is complete and about to be read): everything queued so far
    is issued, and the check that every buffer became a ``.grad`` waits for the end of the pass, when the
    accumulation nodes of the graph have all run."""
    rows = _deferred_rows
    checks = _pending_checks
    if rows:
        flat, keep, dev = [], [], rows[0][1]
        for item, _, part, check in rows:
            flat += item
            keep.append(part)
            checks.append(check)
        count = len(rows)
        del rows[:]
        _lib.call("hivt_reduce_rows_group", (ctypes.c_int64 * len(flat))(*flat), count,
                  torch._C._cuda_getCurrentRawStream(dev.index))
        del keep
    queue = _deferred
    if queue:
        flat, keep, dev = [], [], queue[0][1]
        for item, _, tensors, check in queue:
            flat += item
            keep.append(tensors)
            checks.append(check)
        count = len(queue)
        del queue[:]
        _launch_wgrad_group(flat, count, dev)
        del keep  # the operands stayed alive until the call that reads them was issued
    if final:
        entries = [c for c in checks if c is not None]
        del checks[:]
        _check_adopted(entries)


def _all_leaves(*tensors):
    """The given parameters as a tuple if every one of them is a leaf of the autograd graph (a parameter, not a
    slice or a concatenation of parameters: only then does nothing but the optimizer read its gradient), else
    None.  (None entries -- a layer without a bias -- are skipped.)"""
    params = tuple(t for t in tensors if t is not None)
    return params if all(t.is_leaf for t in params) else None


def _may_defer(leaves) -> bool:
    """Whether a product whose results go to the leaf parameters ``leaves`` may be deferred now: inside
    ``deferred_weight_grads()``, every parameter without a gradient and without hooks."""
    if not leaves or _deferred is None:
        return False
    for p in leaves:
        if p.grad is not None or p._backward_hooks:
            return False
        ready_hooks = getattr(p, "_post_accumulate_grad_hooks", None)
        if ready_hooks and any(h not in _trusted_hooks for h in ready_hooks.values()):
            return False
    return True


def _schedule_flush() -> None:
    if not _deferred and not _deferred_rows:
        torch.autograd.Variable._execution_engine.queue_callback(flush_weight_grads)


def _weight_grad(g: torch.Tensor, g_ld: int, x: torch.Tensor, x_ld: int, m: int, n: int, k: int, dev,
                 bias: bool = True, leaves=None):
    """``(dW [n, k], db [n] or None)`` of ``y = x W^T + b`` from ``g = dL/dy`` [m, n] (fp32, or bfloat16 while
    ``_half_gradients()``) and ``x`` [m, k].  ``leaves``: the leaf parameters the results go to (``_all_leaves``),
    which allows the product to be deferred."""
    half = g.dtype == torch.bfloat16
    dw = torch.empty(n, k, dtype=torch.float32, device=dev)
    db = torch.empty(n, dtype=torch.float32, device=dev) if bias else None
    if m > 0 and _may_defer(leaves):
        _schedule_flush()
        # addresses, not tensors, of the results: autograd takes a gradient over only while it holds the one
        # reference to it
        _deferred.append(([g.data_ptr(), g_ld, x.data_ptr(), x_ld, m, n, k, dw.data_ptr(),
                           db.data_ptr() if bias else 0, 4 if half else 0], dev, (g, x),
                          ((dw.data_ptr(), db.data_ptr() if bias else 0), leaves)))
        return dw, db
    ws = _workspace(_lib.query("hivt_linear_wgrad_workspace_bytes", m, n, k), dev)
    if half:
        _lib.call("hivt_linear_wgrad_half", _p(g), g_ld, _p(x), x_ld, m, n, k, _p(dw), _p(db), _p(ws), ws.numel(),
                  HALF_DY, _stream(g))
        return dw, db
    _lib.call("hivt_linear_wgrad", _p(g), g_ld, _p(x), x_ld, m, n, k, _p(dw), _p(db), _p(ws), ws.numel(), _stream(g))
    return dw, db


_deferred_rows: Optional[list] = None  # LayerNorm partial sums waiting for their one reduction (same scope)


def _ln_backward(dy2, z, z_ld, stats, gamma, beta, relu: int, m: int, n: int, add2, dz, stream, leaves: bool):
    """``hivt_ln_bwd`` (``hivt_ln_bwd_half`` for a bfloat16 ``dz``): fills ``dz`` and returns ``(dgamma, dbeta)``.  With ``leaves`` (gamma and beta are leaves of
    the graph) inside ``deferred_weight_grads()`` the kernel's partial sums stay in a buffer of their own and one
    grouped reduction at the end of the backward pass fills the two gradients of every LayerNorm."""
    dev = dz.device
    dgamma = torch.empty(n, dtype=torch.float32, device=dev)
    dbeta = torch.empty(n, dtype=torch.float32, device=dev)
    add_ld = add2.stride(0) if add2 is not None else 0
    name = "hivt_ln_bwd_half" if dz.dtype == torch.bfloat16 else "hivt_ln_bwd"
    if m > 0 and _may_defer(leaves):
        rows = _lib.query("hivt_ln_bwd_partial_rows", m)
        part = torch.empty(_lib.query("hivt_ln_bwd_workspace_bytes", m, n) // 4, dtype=torch.float32, device=dev)
        _lib.call(name, _p(dy2), dy2.stride(0), _p(z), z_ld, _p(stats), _p(gamma), _p(beta), int(relu), m, n,
                  _p(add2), add_ld, _p(dz), dz.stride(0), None, None, _p(part), part.numel() * 4, stream)
        _schedule_flush()
        _deferred_rows.append(([part.data_ptr(), rows, 2 * n, 2 * n, dgamma.data_ptr(), n, dbeta.data_ptr(), 0], dev,
                               part, ((dgamma.data_ptr(), dbeta.data_ptr()), leaves)))
        return dgamma, dbeta
    ws = _workspace(_lib.query("hivt_ln_bwd_workspace_bytes", m, n), dev)
    _lib.call(name, _p(dy2), dy2.stride(0), _p(z), z_ld, _p(stats), _p(gamma), _p(beta), int(relu), m, n,
              _p(add2), add_ld, _p(dz), dz.stride(0), _p(dgamma), _p(dbeta), _p(ws), ws.numel(), stream)
    return dgamma, dbeta


class _FusedLinear(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, weight, bias, add, gamma, beta, relu, eps, add_index, dropout_p, seed, residual,
                weight_b=None, bias_b=None, stacked=None):
        _need_gpu(x, weight)
        lead = x.shape[:-1]
        x2 = _rows2(x)
        m, k = x2.shape
        n = weight.shape[0]
        # two layers on two column blocks of x whose outputs are summed: one product with the weights side by side
        # and the biases added up; their gradients go to the two layers' own parameters (backward)
        ctx.pair = None
        if weight_b is not None:
            if (bias is None) != (bias_b is None) or weight_b.shape[0] != n:
                raise ValueError("the two layers of a pair have the same output width and both or neither a bias")
            ctx.pair = (int(weight.shape[1]), _all_leaves(weight, bias), _all_leaves(weight_b, bias_b))
            if stacked is not None:  # ([weight | weight_b], bias + bias_b) made by stack_pairs for this pass
                weight, bias = stacked
                if weight.shape != (n, ctx.pair[0] + weight_b.shape[1]):
                    raise ValueError("the stacked weight does not match the two layers")
            else:
                weight = torch.cat([weight, weight_b], dim=1)
                bias = bias + bias_b if bias is not None else None
        if weight.shape[1] != k:
            raise ValueError(f"weight {tuple(weight.shape)} does not match input width {k}")
        w2 = _rows2(weight)
        ln = gamma is not None
        add2 = _rows2(add) if add is not None else None
        if add_index is not None:
            if add is None or add_index.dtype != torch.int32 or add_index.shape != (m,):
                raise ValueError("add_index must be an int32 vector with one entry per row, next to a table in `add`")
            add_index = add_index.contiguous()
        res2 = None
        if residual is not None:
            res2 = _rows2(residual)
            if res2.shape != (m, n):
                raise ValueError(f"residual {tuple(residual.shape)} does not match the output [{m}, {n}]")
        if ln and (dropout_p > 0.0 or residual is not None):
            raise ValueError("dropout / residual are available for the plain epilogues only")
        if relu and residual is not None:
            raise ValueError("a residual after a ReLU epilogue is not supported (the ReLU mask is read off the output)")
        y = torch.empty(m, n, dtype=torch.float32, device=x.device)
        z = torch.empty(m, n, dtype=torch.float32, device=x.device) if ln else None
        stats = torch.empty(m, 2, dtype=torch.float32, device=x.device) if ln else None
        epi = (EPI_LN_RELU if relu else EPI_LN) if ln else (EPI_RELU if relu else EPI_NONE)
        gamma_c = gamma.contiguous() if ln else None
        beta_c = beta.contiguous() if ln else None
        bias_c = bias.contiguous() if bias is not None else None
        _lib.call("hivt_linear_fwd", _p(x2), x2.stride(0), _p(w2), w2.stride(0), _p(bias_c), _p(add2),
                  add2.stride(0) if add2 is not None else 0, _p(add_index),
                  add2.shape[0] if add_index is not None else 0, _p(gamma_c), _p(beta_c), float(eps), epi, 1.0,
                  float(dropout_p), int(seed), _seed_ptr(dropout_p), _p(res2), res2.stride(0) if res2 is not None else 0,
                  m, n, k, _p(y), n, _p(z), n, _p(stats), _stream(x))
        need_y = relu and not ln  # the ReLU (and, with it, the dropout) mask is read off the output
        ctx.save_for_backward(x2, w2, z, stats, gamma_c, beta_c, y if need_y else None, add_index,
                              DEVICE_SEED if dropout_p > 0.0 else None)
        ctx.cfg = (bool(relu), ln, tuple(lead), bias is not None, add is not None,
                   tuple(add.shape) if add is not None else None, float(dropout_p), int(seed),
                   tuple(residual.shape) if residual is not None else None)
        ctx.res_leaf = residual is not None and residual.is_leaf
        ctx.add_leaf = add is not None and add.is_leaf
        ctx.param_leaves = _all_leaves(weight, bias) if ctx.pair is None else None
        ctx.norm_leaves = _all_leaves(gamma, beta) if ln else None
        return y.view(*lead, n)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, w2, z, stats, gamma, beta, y_relu, add_index, seed_dev = ctx.saved_tensors
        relu, ln, lead, has_bias, has_add, add_shape, dropout_p, seed, res_shape = ctx.cfg
        m, k = x2.shape
        n = w2.shape[0]
        dev = x2.device
        dy2 = _rows2(dy)
        s = _stream(x2)
        dgamma = dbeta = None
        # a dz made here that only the two products read (no gradient of `add` to take from it): bfloat16 in
        # memory while those products round it to bfloat16 anyway
        half = (_half_gradients() and n % 4 == 0 and k % 4 == 0 and not (has_add and ctx.needs_input_grad[3])
                and x2.stride(0) % 4 == 0 and w2.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0 and w2.data_ptr() % 16 == 0)
        if ln:
            dz = torch.empty(m, n, dtype=torch.bfloat16 if half else torch.float32, device=dev)
            dgamma, dbeta = _ln_backward(dy2, z, n, stats, gamma, beta, int(relu), m, n, None, dz, s, ctx.norm_leaves)
        elif dropout_p > 0.0:
            half = half and dy2.stride(0) % 4 == 0 and dy2.data_ptr() % 16 == 0
            dz = torch.empty(m, n, dtype=torch.bfloat16 if half else torch.float32, device=dev)
            _lib.call("hivt_dropout_bwd_half" if half else "hivt_dropout_bwd", _p(dy2), dy2.stride(0), _p(y_relu), n,
                      int(relu), dropout_p, seed, _p(seed_dev), m, n, _p(dz), n, s)
        elif relu:
            dz = torch.ops.aten.threshold_backward(dy2, y_relu, 0.0)  # dy where y > 0, one launch
        else:
            dz = dy2
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(m, k, dtype=torch.float32, device=dev)
            if dz.dtype == torch.bfloat16:
                _lib.call("hivt_linear_bwd_data_half", _p(dz), n, _p(w2), w2.stride(0), m, n, k, None, 0, 1.0, None, 0,
                          _p(dx), k, HALF_DY, s)
            else:
                _lib.call("hivt_linear_bwd_data", _p(dz), dz.stride(0), _p(w2), w2.stride(0), m, n, k, None, 0, 1.0, None,
                          0, _p(dx), k, s)
            dx = dx.view(*lead, k)
        dw_b = db_b = None
        if ctx.pair is not None:
            k0, leaves_a, leaves_b = ctx.pair
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                dw, db = _weight_grad(dz, dz.stride(0), x2, x2.stride(0), m, n, k0, dev, bias=has_bias, leaves=leaves_a)
            if ctx.needs_input_grad[12] or (has_bias and ctx.needs_input_grad[13]):
                dw_b, db_b = _weight_grad(dz, dz.stride(0), x2[:, k0:], x2.stride(0), m, n, k - k0, dev, bias=has_bias,
                                          leaves=leaves_b)
        elif ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dw, db = _weight_grad(dz, dz.stride(0), x2, x2.stride(0), m, n, k, dev, bias=has_bias,
                                  leaves=ctx.param_leaves)
        dadd = None
        if has_add and ctx.needs_input_grad[3]:
            if add_index is None:
                dadd = _pass_through_grad(dz, add_shape, ctx.add_leaf)
            else:  # gradient of the table: per-code sums of the rows of dz
                codes = add_shape[0]
                dadd = torch.empty(codes, n, dtype=torch.float32, device=dev)
                ws = _workspace(_lib.query("hivt_rowsum_by_code_workspace_bytes", m, n, codes), dev)
                _lib.call("hivt_rowsum_by_code", _p(dz), dz.stride(0), _p(add_index), m, n, codes, _p(dadd), _p(ws),
                          ws.numel(), s)
        dres = _pass_through_grad(dy, res_shape, ctx.res_leaf) \
            if (res_shape is not None and ctx.needs_input_grad[11]) else None
        return dx, dw, db, dadd, dgamma, dbeta, None, None, None, None, None, dres, dw_b, db_b, None


# --------------------------------------------------------------------------------------------
# Two linear layers on the same input as one product, without copying their parameters
# --------------------------------------------------------------------------------------------
def pack_rows(params) -> None:
    """Re-lays the given parameters (same trailing shape) back to back in one buffer.  Every parameter
    keeps its identity, shape and values -- ``p.data`` becomes a view of the shared buffer -- so
    optimizers, ``state_dict`` and in-place loads see no difference, while ``_stacked`` can hand the
    row-wise concatenation of neighbours to a kernel without a copy.  Moving the module to another
    device or dtype un-packs it again (every parameter gets storage of its own); the consumers then
    fall back to a concatenation."""
    params = list(params)
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        off = 0
        for p in params:
            n = p.numel()
            p.data = flat[off:off + n].view(p.shape)
            off += n


def _stacked(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``[a; b]`` along dim 0: a view when ``b`` starts where ``a`` ends in the same buffer (see
    ``pack_rows``), a concatenation otherwise.  No autograd history either way."""
    a, b = a.detach(), b.detach()
    if (a.is_contiguous() and b.is_contiguous() and a.shape[1:] == b.shape[1:] and a.dtype == b.dtype
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel()):
        return torch.as_strided(a, (a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride())
    return torch.cat([a, b], dim=0)


def _stacked_all(tensors) -> torch.Tensor:
    """``[t0; t1; ...]`` along dim 0 under the rules of ``_stacked``: a view when the tensors lie back to back."""
    tensors = [t.detach() for t in tensors]
    first = tensors[0]
    at = first.storage_offset()
    for t in tensors:
        if not (t.is_contiguous() and t.shape[1:] == first.shape[1:] and t.dtype == first.dtype
                and t.untyped_storage().data_ptr() == first.untyped_storage().data_ptr() and t.storage_offset() == at):
            return torch.cat(tensors, dim=0)
        at += t.numel()
    return torch.as_strided(first, (sum(t.shape[0] for t in tensors),) + tuple(first.shape[1:]), first.stride())


class _PairLinear(torch.autograd.Function):
    """``[x w_a^T + b_a | x w_b^T + b_b]``: one forward product, one data-gradient product and one
    weight-gradient product for two layers that read the same rows."""

    @staticmethod
    def forward(ctx, x, w_a, b_a, w_b, b_b, table=None, index=None, rowptr_by_index=None, perm_by_index=None):
        _need_gpu(x, w_a, w_b)
        lead = x.shape[:-1]
        x2 = _rows2(x)
        m, k = x2.shape
        if w_a.shape[1] != k or w_b.shape[1] != k:
            raise ValueError(f"weights {tuple(w_a.shape)}, {tuple(w_b.shape)} do not match input width {k}")
        w = _stacked(w_a, w_b)
        b = _stacked(b_a, b_b)
        n = w.shape[0]
        y = torch.empty(m, n, dtype=torch.float32, device=x.device)
        if table is not None:
            # row i of the result also receives row index[i] of `table` (the per-source-node terms of an edge's
            # key | value): added where the product's tile is stored, so that no later kernel gathers them
            table = _rows2(table)
            if table.shape[1] != n or index is None or index.shape[0] != m or index.dtype != torch.int32:
                raise ValueError("table rows: need table [R, n_a + n_b] and one int32 row index per input row")
            if table.shape[0] == 0 and m > 0:
                raise ValueError("table rows: empty table")
        _lib.call("hivt_linear_fwd", _p(x2), x2.stride(0), _p(w), w.stride(0), _p(b), _p(table),
                  table.stride(0) if table is not None else 0, _p(index) if table is not None else None,
                  table.shape[0] if table is not None else 0, None, None, 0.0,
                  EPI_NONE, 1.0, 0.0, 0, None, None, 0, m, n, k, _p(y), n, None, 0, None, _stream(x))
        ctx.save_for_backward(x2, w, rowptr_by_index, perm_by_index)
        ctx.cfg = (tuple(lead), int(w_a.shape[0]), table.shape[0] if table is not None else 0)
        ctx.param_leaves = _all_leaves(w_a, b_a, w_b, b_b)
        return y.view(*lead, n)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, w, rowptr_by_index, perm_by_index = ctx.saved_tensors
        lead, n_a, table_rows = ctx.cfg
        m, k = x2.shape
        n = w.shape[0]
        dev = x2.device
        half = dy.dtype == torch.bfloat16  # handed over by _EdgeAttention, never by autograd
        if half and (dy.shape != (m, n) or not dy.is_contiguous()):
            raise RuntimeError("a bfloat16 gradient must be a contiguous [rows, n_a + n_b] tensor")
        dy2 = dy if half else _rows2(dy)
        s = _stream(x2)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(m, k, dtype=torch.float32, device=dev)
            # (a caller that runs this node itself may name rows to add: the input's gradient by its other routes)
            add = getattr(ctx, "dx_add", None)
            if add is not None and (add.shape != (m, k) or add.dtype != torch.float32 or add.stride(1) != 1):
                raise RuntimeError("the rows added to the input gradient must be float32 [rows, width]")
            if half:
                _lib.call("hivt_linear_bwd_data_half", _p(dy2), n, _p(w), w.stride(0), m, n, k, None, 0, 1.0, _p(add),
                          _ld(add) if add is not None else 0, _p(dx), k, HALF_DY, s)
            else:
                _lib.call("hivt_linear_bwd_data", _p(dy2), dy2.stride(0), _p(w), w.stride(0), m, n, k, None, 0, 1.0,
                          _p(add), _ld(add) if add is not None else 0, _p(dx), k, s)
            dx = dx.view(*lead, k)
        dw, db = _weight_grad(dy2, dy2.stride(0), x2, x2.stride(0), m, n, k, dev, leaves=ctx.param_leaves)
        d_table = None
        if table_rows and ctx.needs_input_grad[5]:
            # every table row receives the sum of the gradient rows that picked it, in the fixed order of the
            # grouping by index (no atomics)
            d_table = getattr(ctx, "d_table_out", None)  # (a caller that runs this node itself may say where it goes)
            if d_table is None:
                d_table = torch.empty(table_rows, n, dtype=torch.float32, device=dev)
            _lib.call("hivt_segment_sum_rows_half" if half else "hivt_segment_sum_rows", _p(dy2), dy2.stride(0), n,
                      _p(rowptr_by_index), _p(perm_by_index), table_rows, _p(d_table), _ld(d_table), s)
        return dx, dw[:n_a], db[:n_a], dw[n_a:], db[n_a:], d_table, None, None, None


def pair_linear(x: torch.Tensor, a: torch.nn.Linear, b: torch.nn.Linear, table: Optional[torch.Tensor] = None,
                index: Optional[torch.Tensor] = None, rowptr_by_index: Optional[torch.Tensor] = None,
                perm_by_index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[a(x) | b(x)]`` as one product (GPU float32 tensors; both layers with a bias).  With ``table`` [R, n_a + n_b]
    and ``index`` (int32, one entry per row of ``x``) row i of the result is ``[a(x_i) | b(x_i)] + table[index[i]]``;
    the table's gradient needs the rows grouped by index (``rowptr_by_index`` [R + 1], ``perm_by_index``: grouped
    position -> row)."""
    if table is None:
        return _PairLinear.apply(x, a.weight, a.bias, b.weight, b.bias)
    if table.requires_grad and (rowptr_by_index is None or perm_by_index is None):
        raise ValueError("gradient of the table rows needs the grouping of the rows by index")
    return _PairLinear.apply(x, a.weight, a.bias, b.weight, b.bias, table, index, rowptr_by_index, perm_by_index)


class _StandInContext:
    """What a Function's forward / backward use of their context, for a node that runs another node's two halves
    itself: tensors are kept by the outer node's context and handed back here."""

    def __init__(self, needs_input_grad=()):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class _EdgeAttention(torch.autograd.Function):
    """``seg_attn(q, [a(x) | b(x)], rowptr)``: the key | value projection of the edge rows and the attention over
    the CSR rows (``models/local_encoder.py:166-181, 356-371``) as ONE node.  Same launches as ``pair_linear``
    followed by ``seg_attn``; the gradient of the projected rows stays inside the node, where its only readers
    are the projection's two gradient products -- while those run in bf16 mode it is written and read as
    bfloat16 (between two nodes autograd would widen it to the forward tensor's dtype)."""

    @staticmethod
    def forward(ctx, q, x, w_a, b_a, w_b, b_b, rowptr, num_heads, dropout_p, seed, tag, padded, row_order=None):
        proj, attn = _StandInContext(), _StandInContext()
        kv = _PairLinear.forward(proj, x, w_a, b_a, w_b, b_b)
        out = _SegAttn.forward(attn, q, kv, None, rowptr, None, None, None, num_heads, dropout_p, seed, tag, padded,
                               False, row_order)
        ctx.save_for_backward(*proj.saved_tensors, *attn.saved_tensors)
        ctx.split = len(proj.saved_tensors)
        ctx.stages = (proj.cfg, proj.param_leaves, attn.cfg)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out):
        saved = ctx.saved_tensors
        proj_cfg, param_leaves, attn_cfg = ctx.stages
        need = ctx.needs_input_grad
        attn = _StandInContext((need[0], True, False) + (False,) * 10)
        attn.saved_tensors, attn.cfg = saved[ctx.split:], attn_cfg
        proj = _StandInContext((need[1],) + (True,) * 4 + (False,) * 4)
        proj.saved_tensors, proj.cfg, proj.param_leaves = saved[:ctx.split], proj_cfg, param_leaves
        x2, w = saved[0], saved[1]
        k, n = x2.shape[1], w.shape[0]
        attn.dkv_half = (_half_gradients() and k % 4 == 0 and n % 8 == 0 and x2.stride(0) % 4 == 0
                         and w.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)
        grads = _SegAttn.backward(attn, d_out)
        dq, dkv = grads[0], grads[1]
        dx, dw_a, db_a, dw_b, db_b = _PairLinear.backward(proj, dkv)[:5]
        return dq, dx, dw_a, db_a, dw_b, db_b, None, None, None, None, None, None, None


def edge_attention(q: torch.Tensor, x: torch.Tensor, a: torch.nn.Linear, b: torch.nn.Linear, rowptr: torch.Tensor,
                   num_heads: int, dropout_p: float = 0.0, training: bool = False, tag: str = "",
                   padded: bool = False, row_order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``seg_attn(q, pair_linear(x, a, b), rowptr, ...)`` as one autograd node: ``x`` [E, K] are the edge rows in CSR
    order, ``a`` / ``b`` the key and value projections, ``q`` [N, D] the queries.  Results equal the two calls' bit
    for bit in fp32 mode; in "bf16" mode the projected rows' gradient is kept as bfloat16 (the gradient products
    see the same values, the two bias gradients become sums of rounded values)."""
    if x.dim() != 2:
        raise ValueError(f"edge rows must be [E, K], got {tuple(x.shape)}")
    p = float(dropout_p) if training else 0.0
    return _EdgeAttention.apply(q, x, a.weight, a.bias, b.weight, b.bias, rowptr, int(num_heads), p,
                                _fresh_seed() if p > 0.0 else 0, tag, bool(padded), row_order)


class _LayerNorm(torch.autograd.Function):
    """``(y, x_again)``: the normalised rows and the input handed back as a second output.  A pre-norm
    block ``x + block(norm(x))`` takes its residual operand from ``x_again``; autograd then delivers
    the gradients of both routes to ONE backward call, which sums them inside the LayerNorm-backward
    kernel instead of in a separate pass over the tensor."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, last=0):
        _need_gpu(x, gamma, beta)
        lead = x.shape[:-1]
        x2 = _rows2(x)
        m, n = x2.shape
        gamma_c, beta_c = gamma.contiguous(), beta.contiguous()
        y = torch.empty(m, n, dtype=torch.float32, device=x.device)
        stats = torch.empty(m, 2, dtype=torch.float32, device=x.device)
        _lib.call("hivt_ln_fwd", _p(x2), x2.stride(0), _p(gamma_c), _p(beta_c), float(eps), m, n, _p(y), n, _p(stats),
                  _stream(x))
        ctx.save_for_backward(x2, stats, gamma_c, beta_c)
        ctx.lead = tuple(lead)
        ctx.last = int(last)
        ctx.norm_leaves = _all_leaves(gamma, beta)
        ctx.set_materialize_grads(False)
        if last:  # only the last `last` rows are handed back (a residual that exists for those rows alone)
            if x2.data_ptr() != x.data_ptr() or not 0 < last <= m:
                raise ValueError("layer_norm_last: the input must be contiguous and hold at least `last` rows")
            return y.view(*lead, n), x.view(m, n)[m - last:]
        return y.view(*lead, n), x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, dres):
        x2, stats, gamma, beta = ctx.saved_tensors
        m, n = x2.shape
        dev = x2.device
        if dy is None:  # only the pass-through output was used
            if ctx.last:
                raise RuntimeError("layer_norm_last: the normalised rows must take part in the graph")
            return dres, None, None, None, None
        dy2 = _rows2(dy)
        add2 = _rows2(dres) if (dres is not None and not ctx.last) else None
        dx = torch.empty(m, n, dtype=torch.float32, device=dev)
        dgamma, dbeta = _ln_backward(dy2, x2, x2.stride(0), stats, gamma, beta, 0, m, n, add2, dx, _stream(x2),
                                     ctx.norm_leaves)
        if ctx.last and dres is not None:  # the residual route's gradient reaches the last rows only
            dx[m - ctx.last:].add_(_rows2(dres))
        return dx.view(*ctx.lead, n), dgamma, dbeta, None, None


def layer_norm(x: torch.Tensor, norm: torch.nn.LayerNorm) -> torch.Tensor:
    """``norm(x)`` over the last dimension on the streaming LayerNorm kernels (width <= 128).  CPU tensors
    (and empty ones) go to the framework's implementation; a GPU tensor the kernels are not built for
    raises ``UnsupportedShapeError``."""
    if not x.is_cuda or x.numel() == 0:
        return norm(x)
    if x.shape[-1] > 128 or x.dtype != torch.float32:
        raise _unsupported(f"LayerNorm over {x.shape[-1]} columns of {x.dtype} (float32, at most 128)")
    return _LayerNorm.apply(x, norm.weight, norm.bias, norm.eps)[0]


def layer_norm_residual(x: torch.Tensor, norm: torch.nn.LayerNorm) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(norm(x), x)`` for a pre-norm block ``x + block(norm(x))``: feed the first to the block and use the
    second as the residual operand, and the backward pass sums the two gradient routes inside the
    LayerNorm-backward kernel."""
    if not x.is_cuda or x.numel() == 0:
        return norm(x), x
    if x.shape[-1] > 128 or x.dtype != torch.float32:
        raise _unsupported(f"LayerNorm over {x.shape[-1]} columns of {x.dtype} (float32, at most 128)")
    return _LayerNorm.apply(x, norm.weight, norm.bias, norm.eps)


def layer_norm_last(x: torch.Tensor, norm: torch.nn.LayerNorm, last: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(norm(x), rows)`` with ``rows`` the last ``last`` rows of ``x`` flattened to [rows, width] -- for a block that
    uses the whole normalised tensor but adds its residual to those rows alone (the temporal encoder's last layer):
    the residual's gradient joins the LayerNorm's inside one node instead of through a zero-filled full-size tensor."""
    if x.shape[-1] > 128 or x.dtype != torch.float32:
        raise _unsupported(f"LayerNorm over {x.shape[-1]} columns of {x.dtype} (float32, at most 128)")
    return _LayerNorm.apply(x.contiguous(), norm.weight, norm.bias, norm.eps, int(last))


class _LastRowQKV(torch.autograd.Function):
    """``q = xn[-rows:] w[:D]^T + b[:D]`` and ``kv = xn w[D:]^T + b[D:]`` for the fused projection ``w`` [3D, D] of an
    attention layer whose queries exist for the last ``rows`` rows only: two forward launches; backward writes the
    key | value route's input gradient, adds the query route's into its last rows in the same launch that computes
    it, and fills ONE gradient for ``w`` and ``b`` (no slices of parameters in the graph, no zero-filled tensors)."""

    @staticmethod
    def forward(ctx, xn, w, b, rows):
        _need_gpu(xn, w, b)
        x2 = _rows2(xn)
        m, d = x2.shape
        if w.shape != (3 * d, d) or b.shape != (3 * d,) or not w.is_contiguous() or not 0 < rows <= m:
            raise ValueError("last_row_qkv: w must be a contiguous [3D, D] matrix, b [3D]")
        b = b.contiguous()
        x_last = x2[m - rows:]
        q = fused_linear(x_last, w[:d], b[:d])
        kv = fused_linear(x2, w[d:], b[d:])
        ctx.save_for_backward(x2, w)
        ctx.rows = int(rows)
        ctx.shape = tuple(xn.shape)
        ctx.leaves = _all_leaves(w, b)
        return q, kv

    @staticmethod
    @once_differentiable
    def backward(ctx, dq, dkv):
        x2, w = ctx.saved_tensors
        m, d = x2.shape
        rows, dev, s = ctx.rows, x2.device, _stream(x2)
        dq, dkv = _rows2(dq), _rows2(dkv)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(m, d, dtype=torch.float32, device=dev)
            w_kv, tail = w[d:], dx[m - rows:]
            _lib.call("hivt_linear_bwd_data", _p(dkv), dkv.stride(0), _p(w_kv), w.stride(0), m, 2 * d, d, None, 0, 1.0,
                      None, 0, _p(dx), d, s)
            _lib.call("hivt_linear_bwd_data", _p(dq), dq.stride(0), _p(w), w.stride(0), rows, d, d, None, 0, 1.0,
                      _p(tail), d, _p(tail), d, s)  # (reads and writes the same rows: += in place)
        dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw = torch.empty(3 * d, d, dtype=torch.float32, device=dev)
            db = torch.empty(3 * d, dtype=torch.float32, device=dev)
            x_last = x2[m - rows:]
            terms = [(dq, x_last, rows, d, dw.data_ptr(), db.data_ptr()),
                     (dkv, x2, m, 2 * d, dw.data_ptr() + 4 * d * d, db.data_ptr() + 4 * d)]
            if _may_defer(ctx.leaves):
                for i, (g, x, mm, n, dw_at, db_at) in enumerate(terms):
                    _schedule_flush()
                    # (the buffers are adopted whole: the check looks at the first term's addresses, the gradient's own)
                    _deferred.append(([g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), mm, n, d, dw_at, db_at, 0], dev,
                                      (g, x), ((dw_at, db_at) if i == 0 else (), ctx.leaves)))
            else:
                flat = []
                for g, x, mm, n, dw_at, db_at in terms:
                    flat += [g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), mm, n, d, dw_at, db_at, 0]
                _launch_wgrad_group(flat, len(terms), dev)
        return (dx.view(ctx.shape) if dx is not None else None), dw, db, None


def last_row_qkv(xn: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, rows: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Queries of the last ``rows`` rows and key | value rows of all rows of ``xn`` (flattened to [M, D]) under the
    fused projection ``weight`` [3D, D]: ``(q [rows, D], kv [M, 2D])``."""
    return _LastRowQKV.apply(xn, weight, bias, int(rows))


def fused_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                 add: Optional[torch.Tensor] = None, layer_norm: Optional[torch.nn.LayerNorm] = None,
                 relu: bool = False, add_index: Optional[torch.Tensor] = None, dropout_p: float = 0.0,
                 training: bool = False, residual: Optional[torch.Tensor] = None,
                 weight_b: Optional[torch.Tensor] = None, bias_b: Optional[torch.Tensor] = None,
                 stacked: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """``[residual +] [dropout]([relu]([layer_norm](x @ weight.T + bias [+ add])))`` in one launch (exact
    fp32 on the matrix cores).  With ``add_index`` (int32, one entry per row) ``add`` is a small table
    [codes <= 16, N] and row m adds ``add[add_index[m]]``.  Dropout (active when ``training``) and the
    residual are part of the plain epilogues only (no ``layer_norm``).  With ``weight_b`` / ``bias_b`` the product is
    ``x[:, :k0] @ weight.T + x[:, k0:] @ weight_b.T + bias + bias_b`` (two layers whose outputs are summed, as one
    product over the weights side by side; ``stacked``: that matrix and the summed bias if ``stack_pairs`` has
    made them already).

    CPU tensors are refused; a LayerNorm epilogue wider than 128 columns raises ``UnsupportedShapeError``."""
    _need_gpu(x, weight)
    n = weight.shape[0]
    p = float(dropout_p) if training else 0.0
    if not linear_fused_supported(n, layer_norm is not None):
        raise _unsupported(f"a Linear + LayerNorm epilogue over {n} columns (at most 128)")
    gamma = layer_norm.weight if layer_norm is not None else None
    beta = layer_norm.bias if layer_norm is not None else None
    eps = layer_norm.eps if layer_norm is not None else 0.0
    return _FusedLinear.apply(x, weight, bias, add, gamma, beta, bool(relu), eps, add_index, p,
                              _fresh_seed() if p > 0.0 else 0, residual, weight_b, bias_b, stacked)


def stack_pairs(pairs):
    """``[(cat([w_a, w_b], dim=1), b_a + b_b), ...]`` for pairs ``(w_a, b_a, w_b, b_b)`` of float32 GPU layers with
    the same number of rows: all pairs in ONE launch (no autograd history: the results are values for
    ``fused_linear(..., stacked=)``, whose gradients go to the layers' own parameters)."""
    out, flat, keep = [], [], []
    for w_a, b_a, w_b, b_b in pairs:
        _need_gpu(w_a, w_b)
        w_a, w_b = w_a.detach().contiguous(), w_b.detach().contiguous()
        if w_a.dim() != 2 or w_b.dim() != 2 or w_a.shape[0] != w_b.shape[0] or (b_a is None) != (b_b is None):
            raise ValueError("stack_pairs: two [N, K] weights with the same N, both or neither with a bias")
        n = w_a.shape[0]
        w = torch.empty(n, w_a.shape[1] + w_b.shape[1], dtype=torch.float32, device=w_a.device)
        b = None
        if b_a is not None:
            b_a, b_b = b_a.detach().contiguous(), b_b.detach().contiguous()
            b = torch.empty(n, dtype=torch.float32, device=w_a.device)
        flat += [w_a.data_ptr(), w_b.data_ptr(), w.data_ptr(), n, w_a.shape[1], w_b.shape[1],
                 b_a.data_ptr() if b is not None else 0, b_b.data_ptr() if b is not None else 0,
                 b.data_ptr() if b is not None else 0]
        keep.append((w_a, w_b, b_a, b_b))
        out.append((w, b))
    if out:
        _lib.call("hivt_stack_pairs", (ctypes.c_int64 * len(flat))(*flat), len(out), _stream(out[0][0]))
    return out


# --------------------------------------------------------------------------------------------
# Temporal encoder attention core
# --------------------------------------------------------------------------------------------
class _TemporalAttn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, qkv, num_heads, dropout_p, seed):
        _need_gpu(qkv)
        s_len, n, three_d = qkv.shape
        d_model = three_d // 3
        qkv = qkv.contiguous()
        out = torch.empty(s_len, n, d_model, dtype=torch.float32, device=qkv.device)
        lse = torch.empty(n, num_heads, s_len, dtype=torch.float32, device=qkv.device)
        _lib.call("hivt_temporal_attn_fwd", _p(qkv), s_len, n, d_model, num_heads, float(dropout_p), int(seed),
                  _seed_ptr(dropout_p), _p(out), _p(lse), _stream(qkv))
        ctx.save_for_backward(qkv, out, lse, DEVICE_SEED if dropout_p > 0.0 else None)
        ctx.cfg = (int(num_heads), float(dropout_p), int(seed))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out):
        qkv, out, lse, seed_dev = ctx.saved_tensors
        num_heads, dropout_p, seed = ctx.cfg
        s_len, n, three_d = qkv.shape
        d_out = d_out.contiguous()
        d_qkv = torch.empty_like(qkv)
        _lib.call("hivt_temporal_attn_bwd", _p(d_out), _p(qkv), _p(out), _p(lse), s_len, n, three_d // 3, num_heads,
                  dropout_p, seed, _p(seed_dev), _p(d_qkv), _stream(qkv))
        return d_qkv, None, None, None


def temporal_attention(qkv: torch.Tensor, num_heads: int, dropout_p: float = 0.0, training: bool = False
                       ) -> torch.Tensor:
    """Causal self-attention over dim 0 of the fused projection ``qkv`` [S, N, 3D] -> [S, N, D]."""
    p = float(dropout_p) if training else 0.0
    return _TemporalAttn.apply(qkv, int(num_heads), p, _fresh_seed() if p > 0.0 else 0)


def temporal_attention_supported(seq_len: int, d_model: int, num_heads: int) -> bool:
    return seq_len <= 32 and (d_model, d_model // num_heads) in ((128, 16), (64, 8), (32, 4), (16, 8), (256, 32)) \
        and (num_heads * 32 <= 256 or (num_heads * 32) % 256 == 0)


# --------------------------------------------------------------------------------------------
# First stage of the embedding MLPs (two-component inputs)
# --------------------------------------------------------------------------------------------
def embed2_supported(in_channels, width: int) -> bool:
    return len(in_channels) in (1, 2) and all(c == 2 for c in in_channels) and width in (64, 128)


class _Embed2(torch.autograd.Function):

    @staticmethod
    def forward(ctx, f, eps, *params):
        branches = len(params) // 4
        _need_gpu(f, *params)
        lead = f.shape[:-1]
        f2 = _rows2(f)
        if f2.stride(0) % 2 != 0 or f2.data_ptr() % 8 != 0:
            f2 = f2.contiguous()
        m = f2.shape[0]
        ps = [p.contiguous() for p in params]
        ws, bs, gs, betas = ps[0::4], ps[1::4], ps[2::4], ps[3::4]
        d = ws[0].shape[0]
        h = torch.empty(m, branches * d, dtype=torch.float32, device=f.device)
        _lib.call("hivt_embed2_fwd", _p(f2), f2.stride(0), branches, _ptr_array(ws), _ptr_array(bs), _ptr_array(gs),
                  _ptr_array(betas), float(eps), m, d, _p(h), _stream(f))
        ctx.save_for_backward(f2, *ps)
        ctx.cfg = (branches, d, float(eps), tuple(lead), tuple(f.shape))
        ctx.param_leaves = _all_leaves(*params)
        return h.view(*lead, branches * d)

    @staticmethod
    @once_differentiable
    def backward(ctx, dh):
        f2, *ps = ctx.saved_tensors
        branches, d, eps, lead, f_shape = ctx.cfg
        ws, bs, gs, betas = ps[0::4], ps[1::4], ps[2::4], ps[3::4]
        m = f2.shape[0]
        dev = f2.device
        dh2 = _rows2(dh)
        if dh2.stride(0) != branches * d or dh2.data_ptr() % 16 != 0:
            dh2 = dh2.contiguous()
        dparams = torch.empty(branches, 5 * d, dtype=torch.float32, device=dev)
        df = torch.empty(m, 2 * branches, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        nbytes = _lib.query("hivt_embed2_bwd_workspace_bytes", m, d, branches)
        grads = []
        for b in range(branches):
            row = dparams[b]
            grads += [row[:2 * d].view(d, 2), row[2 * d:3 * d], row[3 * d:4 * d], row[4 * d:5 * d]]
        if m > 0 and _may_defer(ctx.param_leaves):
            # the workgroups' partial sums wait for the one grouped reduction at the end of the backward pass
            rows = _lib.query("hivt_embed2_bwd_partial_rows", m, d, branches)
            part = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            _lib.call("hivt_embed2_bwd", _p(dh2), _p(f2), f2.stride(0), branches, _ptr_array(ws), _ptr_array(bs),
                      _ptr_array(gs), _ptr_array(betas), eps, m, d, None, _p(df), _p(part), nbytes, _stream(f2))
            _schedule_flush()
            cols = branches * 5 * d
            _deferred_rows.append(([part.data_ptr(), rows, cols, cols, dparams.data_ptr(), cols, 0, 0], dev, part,
                                   (tuple(g.data_ptr() for g in grads), ctx.param_leaves)))
        else:
            ws_buf = _workspace(nbytes, dev)
            _lib.call("hivt_embed2_bwd", _p(dh2), _p(f2), f2.stride(0), branches, _ptr_array(ws), _ptr_array(bs),
                      _ptr_array(gs), _ptr_array(betas), eps, m, d, _p(dparams), _p(df), _p(ws_buf), ws_buf.numel(),
                      _stream(f2))
        if df is not None:
            if f_shape[-1] != 2 * branches:  # the input row was wider than the columns the branches read
                full = torch.zeros(m, f_shape[-1], dtype=torch.float32, device=dev)
                full[:, :2 * branches] = df
                df = full
            df = df.view(f_shape)
        return (df, None, *grads)


def embed2(f: torch.Tensor, branch_params, eps: float) -> torch.Tensor:
    """``[ReLU(LN_b(f[..., 2b:2b+2] @ W_b.T + bias_b)) for b in branches]`` side by side, one launch.
    ``branch_params`` is a list (1 or 2 entries) of ``(weight [D,2], bias, ln_weight, ln_bias)``."""
    flat = [t for group in branch_params for t in group]
    if f.shape[-1] < 2 * len(branch_params):
        raise ValueError("input rows are narrower than two components per branch")
    return _Embed2.apply(f, eps, *flat)


# --------------------------------------------------------------------------------------------
# A whole trajectory critic per call
# --------------------------------------------------------------------------------------------
class _Critic(torch.autograd.Function):
    """``(scores [B], ||d score / d x||^2 [B] or None)`` of one critic MLP; forward and backward are one
    library call each."""

    @staticmethod
    def forward(ctx, x, cfg, *params):
        neg_slope, dropout_p, seed, with_r1 = cfg
        weights, biases = [p.contiguous() for p in params[0::2]], [p.contiguous() for p in params[1::2]]
        _need_gpu(x, *weights)
        layers = len(weights) - 1
        x2 = x.reshape(x.shape[0], -1)
        if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < x2.shape[1]) or x2.dtype != torch.float32:
            x2 = x2.contiguous().float()
        b = x2.shape[0]
        widths_list = [x2.shape[1]] + [w.shape[0] for w in weights[:-1]]
        for l, w in enumerate(weights[:-1]):
            if w.shape[1] != widths_list[l]:
                raise ValueError("critic layer widths do not chain")
        if weights[-1].shape != (1, widths_list[-1]):
            raise ValueError("the output layer must be [1, hidden]")
        widths = (ctypes.c_int32 * (layers + 1))(*widths_list)
        dev = x2.device
        saved = torch.empty(_lib.query("hivt_critic_saved_floats", b, layers, tuple(widths_list), int(with_r1)),
                            dtype=torch.float32, device=dev)
        scores = torch.empty(b, dtype=torch.float32, device=dev)
        gn2 = torch.empty(b, dtype=torch.float32, device=dev) if with_r1 else None
        _lib.call("hivt_critic_fwd", _p(x2), x2.stride(0) if b > 0 else widths_list[0], b, layers, widths,
                  _ptr_array(weights), _ptr_array(biases), float(neg_slope), float(dropout_p), int(seed),
                  _seed_ptr(dropout_p), int(with_r1), _p(saved), _p(scores), _p(gn2), _stream(x2))
        ctx.save_for_backward(x2, saved, *weights)
        ctx.cfg = (float(neg_slope), float(dropout_p), widths_list, tuple(x.shape), [tuple(t.shape) for t in biases])
        ctx.set_materialize_grads(False)
        if with_r1:
            return scores, gn2
        return scores

    @staticmethod
    @once_differentiable
    def backward(ctx, g_scores, g_norm2=None):
        x2, saved, *weights = ctx.saved_tensors
        neg_slope, dropout_p, widths_list, x_shape, bias_shapes = ctx.cfg
        layers = len(weights) - 1
        b = x2.shape[0]
        dev = x2.device
        widths = (ctypes.c_int32 * (layers + 1))(*widths_list)
        if not any(ctx.needs_input_grad[2:]):  # frozen critic (the generator's pass): data gradient only
            dx = torch.empty(b, widths_list[0], dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
            if dx is not None:
                ws = _workspace(_lib.query("hivt_critic_workspace_bytes", b, layers, tuple(widths_list)), dev)
                gs = g_scores.contiguous() if g_scores is not None else None
                _lib.call("hivt_critic_bwd", _p(gs), None, _p(x2), x2.stride(0) if b > 0 else widths_list[0], b, layers,
                          widths, _ptr_array(weights), neg_slope, dropout_p, _p(saved), None, None, _p(dx), _p(ws),
                          ws.numel(), _stream(x2))
            return (dx.view(x_shape) if dx is not None else None, None, *([None] * (2 * len(weights))))
        dws = [torch.empty_like(w) for w in weights]
        dbs = [torch.empty(s, dtype=torch.float32, device=dev) for s in bias_shapes]
        dx = torch.empty(b, widths_list[0], dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        ws = _workspace(_lib.query("hivt_critic_workspace_bytes", b, layers, tuple(widths_list)), dev)
        gs = g_scores.contiguous() if g_scores is not None else None
        gn = g_norm2.contiguous() if g_norm2 is not None else None
        _lib.call("hivt_critic_bwd", _p(gs), _p(gn), _p(x2), x2.stride(0) if b > 0 else widths_list[0], b, layers, widths,
                  _ptr_array(weights), neg_slope, dropout_p, _p(saved), _ptr_array(dws), _ptr_array(dbs), _p(dx),
                  _p(ws), ws.numel(), _stream(x2))
        grads = []
        for dw, db in zip(dws, dbs):
            grads += [dw, db]
        return (dx.view(x_shape) if dx is not None else None, None, *grads)


def critic_scores(x: torch.Tensor, weights, biases, neg_slope: float, dropout_p: float, training: bool,
                  with_input_grad_norm: bool = False):
    """Scores [B] of the critic MLP ``Linear -> LeakyReLU -> Dropout`` x L, then a one-wide Linear, over
    ``x`` flattened to [B, -1]; with ``with_input_grad_norm`` also ``||d score_i / d x_i||^2`` [B] as a
    differentiable quantity (what the R1 penalty averages)."""
    if len(weights) != len(biases) or not 2 <= len(weights) <= 9:
        raise ValueError("a critic has 1..8 hidden layers and an output layer")
    p = float(dropout_p) if training else 0.0
    params = [t for pair in zip(weights, biases) for t in pair]
    cfg = (float(neg_slope), p, _fresh_seed() if p > 0.0 else 0, bool(with_input_grad_norm))
    return _Critic.apply(x, cfg, *params)


def _critic_chainable(widths_list, layers: int, b: int, x2: torch.Tensor) -> bool:
    """Whether the library carries a pass through its layer-chain kernel (the route on which passes of one critic
    may share their gradient arrays): the conditions of the C side, restated."""
    return (layers <= 4 and b > 0 and all(w % 4 == 0 for w in widths_list) and x2.stride(0) % 4 == 0
            and x2.data_ptr() % 16 == 0)


class _CriticsMulti(torch.autograd.Function):
    """Several critic passes (all critics, real and generated trajectories) per library call: forward and
    backward are one call each, and the passes share their launches.  ``plan = (inputs, critics, passes)``:
    ``inputs`` input tensors come first, then per critic (``critics[c]`` = its number of linear layers) ``w_1, b_1,
    .., w_o, b_o``; ``passes`` lists ``(input, critic, with_r1, seed)``.  A pass reads the first ``w_1.shape[1]``
    columns of its input's flattened rows.  Outputs per pass: scores [B] and, with ``with_r1``,
    ``||d score / d x||^2`` [B].  In backward the passes of one critic write ONE set of parameter gradients (their
    sum) and the passes over one input ONE input gradient."""

    @staticmethod
    def forward(ctx, cfg, plan, *tensors):
        neg_slope, dropout_p = cfg
        n_inputs, critics, pass_plan = plan[:3]
        ctx.same_as = plan[3] if len(plan) > 3 else None
        xs = tensors[:n_inputs]
        params, at = [], n_inputs
        for n_lin in critics:
            group = tensors[at:at + 2 * n_lin]
            at += 2 * n_lin
            params.append(([p.contiguous() for p in group[0::2]], [p.contiguous() for p in group[1::2]]))
        x2s = []
        for x in xs:
            x2 = x.reshape(x.shape[0], -1)
            if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < x2.shape[1]) or x2.dtype != torch.float32:
                x2 = x2.contiguous().float()
            x2s.append(x2)
        passes, table, keep, outs = [], [], [], []
        for x_idx, c_idx, with_r1, seed in pass_plan:
            weights, biases = params[c_idx]
            x2 = x2s[x_idx]
            _need_gpu(x2, *weights)
            layers = len(weights) - 1
            b = x2.shape[0]
            widths_list = [weights[0].shape[1]] + [w.shape[0] for w in weights[:-1]]
            if widths_list[0] > x2.shape[1]:
                raise ValueError("critic input rows are narrower than the first layer")
            for l, w in enumerate(weights[:-1]):
                if w.shape[1] != widths_list[l]:
                    raise ValueError("critic layer widths do not chain")
            if weights[-1].shape != (1, widths_list[-1]):
                raise ValueError("the output layer must be [1, hidden]")
            dev = x2.device
            widths = (ctypes.c_int32 * (layers + 1))(*widths_list)
            w_ptrs, b_ptrs = _ptr_array(weights), _ptr_array(biases)
            saved = torch.empty(_lib.query("hivt_critic_saved_floats", b, layers, tuple(widths_list), int(with_r1)),
                                dtype=torch.float32, device=dev)
            scores = torch.empty(b, dtype=torch.float32, device=dev)
            gn2 = torch.empty(b, dtype=torch.float32, device=dev) if with_r1 else None
            table += [x2.data_ptr(), x2.stride(0) if b > 0 else widths_list[0], b, layers, ctypes.addressof(widths),
                      ctypes.addressof(w_ptrs), ctypes.addressof(b_ptrs), int(seed), int(with_r1), saved.data_ptr(),
                      scores.data_ptr(), gn2.data_ptr() if with_r1 else 0]
            keep.append((widths, w_ptrs, b_ptrs))
            passes.append((saved, widths_list, bool(with_r1)))
            outs.append(scores)
            if with_r1:
                outs.append(gn2)
        dev = x2s[0].device
        arr = (ctypes.c_int64 * len(table))(*table)
        _lib.call("hivt_critics_fwd", arr, len(passes), float(neg_slope), float(dropout_p), _seed_ptr(dropout_p),
                  torch._C._cuda_getCurrentRawStream(dev.index))
        del keep
        flat = list(x2s) + [saved for saved, _, _ in passes]
        for weights, _ in params:
            flat += weights
        ctx.save_for_backward(*flat)
        ctx.meta = ([tuple(x.shape) for x in xs], [(len(w), [tuple(t.shape) for t in bs]) for w, bs in params],
                    [(x_idx, c_idx, widths_list, with_r1)
                     for (x_idx, c_idx, _, _), (_, widths_list, with_r1) in zip(pass_plan, passes)])
        ctx.cfg = (float(neg_slope), float(dropout_p))
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gouts):
        neg_slope, dropout_p = ctx.cfg
        x_shapes, critic_meta, pass_meta = ctx.meta
        saved_tensors = ctx.saved_tensors
        n_inputs, n_passes = len(x_shapes), len(pass_meta)
        x2s = saved_tensors[:n_inputs]
        saveds = saved_tensors[n_inputs:n_inputs + n_passes]
        weights_of, at = [], n_inputs + n_passes
        for n_lin, _ in critic_meta:
            weights_of.append(list(saved_tensors[at:at + n_lin]))
            at += n_lin
        dev = x2s[0].device
        # which inputs of forward want a gradient (after cfg, plan: the inputs, then every critic's parameters)
        need_x = [ctx.needs_input_grad[2 + i] for i in range(n_inputs)]
        need_params, in_at = [], 2 + n_inputs
        for n_lin, _ in critic_meta:
            need_params.append(any(ctx.needs_input_grad[in_at:in_at + 2 * n_lin]))
            in_at += 2 * n_lin
        # incoming gradients per pass
        grads_in, out_at = [], 0
        for _, _, _, with_r1 in pass_meta:
            grads_in.append((gouts[out_at], gouts[out_at + 1] if with_r1 else None))
            out_at += 2 if with_r1 else 1
        # one set of parameter gradients per critic, shared by its passes when the library can sum them itself
        param_grads = [None] * len(critic_meta)   # (dws, dbs, dw_ptrs, db_ptrs)
        extra_sets = []                           # (critic, dws, dbs): a pass that could not share, added afterwards

        def new_set(c_idx):
            dws = [torch.empty_like(w) for w in weights_of[c_idx]]
            dbs = [torch.empty(shape, dtype=torch.float32, device=dev) for shape in critic_meta[c_idx][1]]
            return dws, dbs, _ptr_array(dws), _ptr_array(dbs)

        table, keep, dx_parts = [], [], [[] for _ in range(n_inputs)]
        for p_idx, (x_idx, c_idx, widths_list, with_r1) in enumerate(pass_meta):
            g_scores, g_norm2 = grads_in[p_idx]
            want_params = need_params[c_idx] and (g_scores is not None or g_norm2 is not None)
            want_x = need_x[x_idx] and g_scores is not None
            if not want_params and not want_x:
                continue
            x2, saved, weights = x2s[x_idx], saveds[p_idx], weights_of[c_idx]
            layers = len(weights) - 1
            b = x2.shape[0]
            gs = g_scores.contiguous() if g_scores is not None else None
            gn = g_norm2.contiguous() if (g_norm2 is not None and want_params) else None
            dx = torch.empty(b, widths_list[0], dtype=torch.float32, device=dev) if want_x else None
            dw_ptrs = db_ptrs = None
            if want_params:
                if param_grads[c_idx] is None:
                    param_grads[c_idx] = new_set(c_idx)
                    _, _, dw_ptrs, db_ptrs = param_grads[c_idx]
                elif _critic_chainable(widths_list, layers, b, x2):
                    _, _, dw_ptrs, db_ptrs = param_grads[c_idx]  # the library adds this pass's sums to the set
                else:
                    dws, dbs, dw_ptrs, db_ptrs = new_set(c_idx)
                    extra_sets.append((c_idx, dws, dbs))
            widths = (ctypes.c_int32 * (layers + 1))(*widths_list)
            w_ptrs = _ptr_array(weights)
            table += [gs.data_ptr() if gs is not None else 0, gn.data_ptr() if gn is not None else 0,
                      x2.data_ptr(), x2.stride(0) if b > 0 else widths_list[0], b, layers, ctypes.addressof(widths),
                      ctypes.addressof(w_ptrs), saved.data_ptr(), ctypes.addressof(dw_ptrs) if dw_ptrs is not None else 0,
                      ctypes.addressof(db_ptrs) if db_ptrs is not None else 0, dx.data_ptr() if dx is not None else 0]
            keep.append((widths, w_ptrs, gs, gn))
            if dx is not None:
                dx_parts[x_idx].append((dx, widths_list[0]))
        # a critic whose first pass cannot take the chain route must not share at all: its later passes were given
        # the same arrays above only if they are chainable, which the first pass then is as well (same widths, same
        # batch); passes over different batch sizes fall under the library's own check
        stream = torch._C._cuda_getCurrentRawStream(dev.index)
        if table:
            count = len(table) // 12
            arr = (ctypes.c_int64 * len(table))(*table)
            ws = _workspace(_lib.value("hivt_critics_bwd_workspace_bytes", arr, count), dev)
            _lib.call("hivt_critics_bwd", arr, count, neg_slope, dropout_p, _p(ws), ws.numel(), stream)
        del keep
        for c_idx, dws, dbs in extra_sets:
            torch._foreach_add_(param_grads[c_idx][0] + param_grads[c_idx][1], dws + dbs)
        grads = [None, None]
        for i in range(n_inputs):
            parts = dx_parts[i]
            if not parts:
                grads.append(None)
                continue
            width = x2s[i].shape[1]
            if len(parts) == 1 and parts[0][1] == width:
                grads.append(parts[0][0].view(x_shapes[i]))
                continue
            total = None
            for first in range(0, len(parts), 4 if len(parts) <= 4 else 3):
                chunk = parts[first:first + (4 if len(parts) <= 4 else 3)]
                if total is not None:
                    chunk = [(total, width)] + chunk
                out = torch.empty(x2s[i].shape[0], width, dtype=torch.float32, device=dev)
                _lib.call("hivt_sum_padded_rows", _ptr_array([t for t, _ in chunk]),
                          (ctypes.c_int32 * len(chunk))(*[w for _, w in chunk]), len(chunk), x2s[i].shape[0], width,
                          _p(out), stream)
                total = out
            grads.append(total.view(x_shapes[i]))
        param_at = len(grads)
        for c_idx, (n_lin, _) in enumerate(critic_meta):
            if param_grads[c_idx] is None:
                grads += [None] * (2 * n_lin)
            else:
                dws, dbs, _, _ = param_grads[c_idx]
                for dw, db in zip(dws, dbs):
                    grads += [dw, db]
        if ctx.same_as is not None:
            into, what = [], []
            for j, first in enumerate(ctx.same_as):
                if first == j or grads[param_at + j] is None:
                    continue
                if grads[param_at + first] is None:
                    grads[param_at + first] = grads[param_at + j]
                else:
                    into.append(grads[param_at + first])
                    what.append(grads[param_at + j])
                grads[param_at + j] = None
            if into:  # one multi-tensor launch for all shared parameters
                torch._foreach_add_(into, what)
        return tuple(grads)


def critic_scores_multi(passes, neg_slope: float, dropout_p: float, training: bool):
    """``passes``: tuples ``(x, weights, biases, with_input_grad_norm)``, one per critic pass (the layout of
    ``critic_scores``; all passes share the LeakyReLU slope and the dropout rate).  ``x`` may hold more columns per
    flattened row than the critic's first layer reads (a longer trajectory than the critic's horizon): the pass
    takes the leading ones.  Passes that name the same weight tensors are passes of ONE critic (their parameter
    gradients come out summed), passes that name the same ``x`` share its gradient.  Returns per pass ``scores`` or
    ``(scores, grad_norm2)``; all passes run in one library call forward and one backward."""
    p = float(dropout_p) if training else 0.0
    xs, x_index, critics, critic_index, plan, tensors = [], {}, [], {}, [], []
    for x, weights, biases, with_r1 in passes:
        if len(weights) != len(biases) or not 2 <= len(weights) <= 9:
            raise ValueError("a critic has 1..8 hidden layers and an output layer")
        if id(x) not in x_index:
            x_index[id(x)] = len(xs)
            xs.append(x)
        key = tuple(id(t) for pair in zip(weights, biases) for t in pair)
        if key not in critic_index:
            critic_index[key] = len(critics)
            critics.append(len(weights))
            tensors += [t for pair in zip(weights, biases) for t in pair]
        plan.append((x_index[id(x)], critic_index[key], bool(with_r1), _fresh_seed() if p > 0.0 else 0))
    # a parameter tensor that two critics of the call share (the biases of one critic under two normalisations of
    # its weights): its gradient comes out once, summed, at its first place
    first_at, same_as = {}, []
    for j, t in enumerate(tensors):
        same_as.append(first_at.setdefault(id(t), j))
    outs = _CriticsMulti.apply((float(neg_slope), p), (len(xs), tuple(critics), tuple(plan), tuple(same_as)), *xs, *tensors)
    result, at = [], 0
    for _, _, _, with_r1 in passes:
        if with_r1:
            result.append((outs[at], outs[at + 1]))
            at += 2
        else:
            result.append(outs[at])
            at += 1
    return result


# --------------------------------------------------------------------------------------------
# Pre-norm feed-forward block as ONE autograd node
# --------------------------------------------------------------------------------------------
class _PreNormFeedForward(torch.autograd.Function):
    """``x + drop2(lin2(drop1(relu(lin1(norm(x))))))`` (``models/local_encoder.py:153,201-202,267,278-280``;
    ``models/global_interactor.py:105``): three launches forward; backward carries the gradient through
    the ReLU / first dropout inside the second Linear's data-gradient product and sums the residual route
    inside the LayerNorm backward, so no [M, 4D] or [M, D] tensor is passed over on its own."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, w1, b1, w2, b2, p1, p2, seed1, seed2):
        _need_gpu(x, w1, w2)
        lead = x.shape[:-1]
        x2 = _rows2(x)
        m, d = x2.shape
        hdim = w1.shape[0]
        if w1.shape[1] != d or w2.shape != (d, hdim):
            raise ValueError(f"feed-forward weights {tuple(w1.shape)}, {tuple(w2.shape)} do not fit rows of width {d}")
        dev = x.device
        gamma_c, beta_c = gamma.contiguous(), beta.contiguous()
        w1c, w2c, b1c, b2c = _rows2(w1), _rows2(w2), b1.contiguous(), b2.contiguous()
        s = _stream(x)
        xn = torch.empty(m, d, dtype=torch.float32, device=dev)
        stats = torch.empty(m, 2, dtype=torch.float32, device=dev)
        h = torch.empty(m, hdim, dtype=torch.float32, device=dev)
        y = torch.empty(m, d, dtype=torch.float32, device=dev)
        # while the gradient products run in bf16 mode, the gradient through ReLU / dropout reads one bit per hidden
        # element (written here) instead of the hidden rows themselves
        bits = torch.empty(m, (hdim + 31) // 32, dtype=torch.int32, device=dev) if _half_gradients() else None
        _lib.call("hivt_ln_fwd", _p(x2), x2.stride(0), _p(gamma_c), _p(beta_c), float(eps), m, d, _p(xn), d, _p(stats), s)
        _lib.call("hivt_linear_fwd", _p(xn), d, _p(w1c), w1c.stride(0), _p(b1c), None, 0, None, 0, None, None, 0.0,
                  EPI_RELU | (EPI_SIGN_BITS if bits is not None else 0), 1.0, float(p1), int(seed1), _seed_ptr(p1), None,
                  0, m, hdim, d, _p(h), hdim, None, 0, _p(bits), s)
        _lib.call("hivt_linear_fwd", _p(h), hdim, _p(w2c), w2c.stride(0), _p(b2c), None, 0, None, 0, None, None, 0.0,
                  EPI_NONE, 1.0, float(p2), int(seed2), _seed_ptr(p2), _p(x2), x2.stride(0), m, d, hdim, _p(y), d, None,
                  0, None, s)
        ctx.save_for_backward(x2, stats, gamma_c, beta_c, xn, h, w1c, w2c, DEVICE_SEED if p2 > 0.0 else None, bits)
        ctx.cfg = (tuple(lead), float(p1), float(p2), int(seed2))
        ctx.param_leaves = _all_leaves(w1, b1, w2, b2)
        ctx.norm_leaves = _all_leaves(gamma, beta)
        return y.view(*lead, d)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, stats, gamma, beta, xn, h, w1, w2, seed_dev, bits = ctx.saved_tensors
        lead, p1, p2, seed2 = ctx.cfg
        m, d = x2.shape
        hdim = w1.shape[0]
        dev = x2.device
        s = _stream(x2)
        dy2 = _rows2(dy)
        # gradients that only the two bf16 products read: bfloat16 in memory while those are on
        half = _half_gradients() and hdim % 4 == 0 and d % 4 == 0
        half2 = half and p2 > 0.0 and dy2.stride(0) % 4 == 0 and dy2.data_ptr() % 16 == 0
        if half2:
            dz2 = torch.empty(m, d, dtype=torch.bfloat16, device=dev)
            _lib.call("hivt_dropout_bwd_half", _p(dy2), dy2.stride(0), None, 0, 0, p2, seed2, _p(seed_dev), m, d, _p(dz2),
                      d, s)
        elif p2 > 0.0:
            dz2 = torch.empty(m, d, dtype=torch.float32, device=dev)
            _lib.call("hivt_dropout_bwd", _p(dy2), dy2.stride(0), None, 0, 0, p2, seed2, _p(seed_dev), m, d, _p(dz2), d, s)
        else:
            dz2 = dy2
        # second Linear: weight gradient, then the data gradient gated by the ReLU / dropout of its input
        dw2, db2 = _weight_grad(dz2, dz2.stride(0), h, hdim, m, d, hdim, dev, leaves=ctx.param_leaves)
        dz1 = torch.empty(m, hdim, dtype=torch.bfloat16 if half else torch.float32, device=dev)
        dxn = torch.empty(m, d, dtype=torch.float32, device=dev)
        if half and bits is not None:
            _lib.call("hivt_linear_bwd_data_half", _p(dz2), dz2.stride(0), _p(w2), w2.stride(0), m, d, hdim, _p(bits),
                      bits.shape[1], 1.0 / (1.0 - p1), None, 0, _p(dz1), hdim,
                      HALF_DX | MASK_BITS | (HALF_DY if half2 else 0), s)
        elif half:
            _lib.call("hivt_linear_bwd_data_half", _p(dz2), dz2.stride(0), _p(w2), w2.stride(0), m, d, hdim, _p(h), hdim,
                      1.0 / (1.0 - p1), None, 0, _p(dz1), hdim, HALF_DX | (HALF_DY if half2 else 0), s)
        else:
            _lib.call("hivt_linear_bwd_data", _p(dz2), dz2.stride(0), _p(w2), w2.stride(0), m, d, hdim, _p(h), hdim,
                      1.0 / (1.0 - p1), None, 0, _p(dz1), hdim, s)
        # first Linear
        dw1, db1 = _weight_grad(dz1, hdim, xn, d, m, hdim, d, dev, leaves=ctx.param_leaves)
        if half:
            _lib.call("hivt_linear_bwd_data_half", _p(dz1), hdim, _p(w1), w1.stride(0), m, hdim, d, None, 0, 1.0, None, 0,
                      _p(dxn), d, HALF_DY, s)
        else:
            _lib.call("hivt_linear_bwd_data", _p(dz1), hdim, _p(w1), w1.stride(0), m, hdim, d, None, 0, 1.0, None, 0,
                      _p(dxn), d, s)
        # LayerNorm, plus the residual route's gradient (dy itself)
        dx = torch.empty(m, d, dtype=torch.float32, device=dev)
        dgamma, dbeta = _ln_backward(dxn, x2, x2.stride(0), stats, gamma, beta, 0, m, d, dy2, dx, s, ctx.norm_leaves)
        return dx.view(*lead, d), dgamma, dbeta, None, dw1, db1, dw2, db2, None, None, None, None


def pre_norm_feed_forward(x: torch.Tensor, norm: torch.nn.LayerNorm, lin1: torch.nn.Linear, p1: float,
                          lin2: torch.nn.Linear, p2: float, training: bool) -> torch.Tensor:
    """``x + dropout_p2(lin2(dropout_p1(relu(lin1(norm(x))))))`` as one autograd node (GPU tensors, rows of
    at most 128 columns, both Linears with a bias)."""
    p1 = float(p1) if training else 0.0
    p2 = float(p2) if training else 0.0
    return _PreNormFeedForward.apply(x, norm.weight, norm.bias, norm.eps, lin1.weight, lin1.bias, lin2.weight,
                                     lin2.bias, p1, p2, _fresh_seed() if p1 > 0.0 else 0,
                                     _fresh_seed() if p2 > 0.0 else 0)


def pre_norm_feed_forward_supported(x: torch.Tensor, lin1: torch.nn.Linear, lin2: torch.nn.Linear) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] <= 128 and x.numel() > 0
            and lin1.bias is not None and lin2.bias is not None)


# --------------------------------------------------------------------------------------------
# Pre-norm causal self-attention block of the temporal encoder as ONE autograd node
# --------------------------------------------------------------------------------------------
class _PreNormTemporalAttention(torch.autograd.Function):
    """``x + drop(out_proj(causal_attention(in_proj(norm(x)))))`` for x [S, N, D]
    (``models/local_encoder.py:266,269-276``): LayerNorm, the fused q|k|v projection, the attention kernel
    and the output projection with dropout and residual, four launches forward and one node in the graph."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, w_in, b_in, w_out, b_out, num_heads, p_attn, p_out, seed_attn, seed_out):
        _need_gpu(x, w_in, w_out)
        s_len, n, d = x.shape
        x2 = _rows2(x)
        m = x2.shape[0]
        dev = x.device
        st = _stream(x)
        gamma_c, beta_c = gamma.contiguous(), beta.contiguous()
        w_in_c, w_out_c, b_in_c, b_out_c = _rows2(w_in), _rows2(w_out), b_in.contiguous(), b_out.contiguous()
        xn = torch.empty(m, d, dtype=torch.float32, device=dev)
        stats = torch.empty(m, 2, dtype=torch.float32, device=dev)
        qkv = torch.empty(m, 3 * d, dtype=torch.float32, device=dev)
        o = torch.empty(m, d, dtype=torch.float32, device=dev)
        lse = torch.empty(n, num_heads, s_len, dtype=torch.float32, device=dev)
        y = torch.empty(m, d, dtype=torch.float32, device=dev)
        _lib.call("hivt_ln_fwd", _p(x2), x2.stride(0), _p(gamma_c), _p(beta_c), float(eps), m, d, _p(xn), d, _p(stats), st)
        _lib.call("hivt_linear_fwd", _p(xn), d, _p(w_in_c), w_in_c.stride(0), _p(b_in_c), None, 0, None, 0, None, None,
                  0.0, EPI_NONE, 1.0, 0.0, 0, None, None, 0, m, 3 * d, d, _p(qkv), 3 * d, None, 0, None, st)
        _lib.call("hivt_temporal_attn_fwd", _p(qkv), s_len, n, d, num_heads, float(p_attn), int(seed_attn),
                  _seed_ptr(p_attn), _p(o), _p(lse), st)
        _lib.call("hivt_linear_fwd", _p(o), d, _p(w_out_c), w_out_c.stride(0), _p(b_out_c), None, 0, None, 0, None, None,
                  0.0, EPI_NONE, 1.0, float(p_out), int(seed_out), _seed_ptr(p_out), _p(x2), x2.stride(0), m, d, d,
                  _p(y), d, None, 0, None, st)
        ctx.save_for_backward(x2, stats, gamma_c, beta_c, xn, qkv, o, lse, w_in_c, w_out_c,
                              DEVICE_SEED if (p_attn > 0.0 or p_out > 0.0) else None)
        ctx.cfg = (s_len, n, int(num_heads), float(p_attn), float(p_out), int(seed_attn), int(seed_out))
        ctx.param_leaves = _all_leaves(w_in, b_in, w_out, b_out)
        ctx.norm_leaves = _all_leaves(gamma, beta)
        return y.view(s_len, n, d)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, stats, gamma, beta, xn, qkv, o, lse, w_in, w_out, seed_dev = ctx.saved_tensors
        s_len, n, num_heads, p_attn, p_out, seed_attn, seed_out = ctx.cfg
        m, d = x2.shape
        dev = x2.device
        st = _stream(x2)
        dy2 = _rows2(dy)
        # gradients that only the bf16 products read (the dropout's, the attention's): bfloat16 in memory
        half = _half_gradients() and d % 4 == 0
        half_z = half and p_out > 0.0 and dy2.stride(0) % 4 == 0 and dy2.data_ptr() % 16 == 0
        if half_z:
            dz = torch.empty(m, d, dtype=torch.bfloat16, device=dev)
            _lib.call("hivt_dropout_bwd_half", _p(dy2), dy2.stride(0), None, 0, 0, p_out, seed_out, _p(seed_dev), m, d,
                      _p(dz), d, st)
        elif p_out > 0.0:
            dz = torch.empty(m, d, dtype=torch.float32, device=dev)
            _lib.call("hivt_dropout_bwd", _p(dy2), dy2.stride(0), None, 0, 0, p_out, seed_out, _p(seed_dev), m, d, _p(dz),
                      d, st)
        else:
            dz = dy2
        dw_out, db_out = _weight_grad(dz, dz.stride(0), o, d, m, d, d, dev, leaves=ctx.param_leaves)
        do = torch.empty(m, d, dtype=torch.float32, device=dev)
        if half_z:
            _lib.call("hivt_linear_bwd_data_half", _p(dz), d, _p(w_out), w_out.stride(0), m, d, d, None, 0, 1.0, None, 0,
                      _p(do), d, HALF_DY, st)
        else:
            _lib.call("hivt_linear_bwd_data", _p(dz), dz.stride(0), _p(w_out), w_out.stride(0), m, d, d, None, 0, 1.0,
                      None, 0, _p(do), d, st)
        dqkv = torch.empty(m, 3 * d, dtype=torch.bfloat16 if half else torch.float32, device=dev)
        _lib.call("hivt_temporal_attn_bwd_half" if half else "hivt_temporal_attn_bwd", _p(do), _p(qkv), _p(o), _p(lse),
                  s_len, n, d, num_heads, p_attn, seed_attn, _p(seed_dev), _p(dqkv), st)
        dw_in, db_in = _weight_grad(dqkv, 3 * d, xn, d, m, 3 * d, d, dev, leaves=ctx.param_leaves)
        dxn = torch.empty(m, d, dtype=torch.float32, device=dev)
        if half:
            _lib.call("hivt_linear_bwd_data_half", _p(dqkv), 3 * d, _p(w_in), w_in.stride(0), m, 3 * d, d, None, 0, 1.0,
                      None, 0, _p(dxn), d, HALF_DY, st)
        else:
            _lib.call("hivt_linear_bwd_data", _p(dqkv), 3 * d, _p(w_in), w_in.stride(0), m, 3 * d, d, None, 0, 1.0, None,
                      0, _p(dxn), d, st)
        dx = torch.empty(m, d, dtype=torch.float32, device=dev)
        dgamma, dbeta = _ln_backward(dxn, x2, x2.stride(0), stats, gamma, beta, 0, m, d, dy2, dx, st, ctx.norm_leaves)
        return (dx.view(s_len, n, d), dgamma, dbeta, None, dw_in, db_in, dw_out, db_out, None, None, None, None, None)


def pre_norm_temporal_attention(x: torch.Tensor, norm: torch.nn.LayerNorm, mha: torch.nn.MultiheadAttention,
                                p_out: float, training: bool) -> torch.Tensor:
    """``x + dropout_p_out(mha.out_proj(causal self-attention(mha.in_proj(norm(x)))))`` over the first
    axis of x [S, N, D] as one autograd node (needs ``temporal_attention_supported`` and D <= 128)."""
    p_attn = float(mha.dropout) if training else 0.0
    p_out = float(p_out) if training else 0.0
    return _PreNormTemporalAttention.apply(x, norm.weight, norm.bias, norm.eps, mha.in_proj_weight, mha.in_proj_bias,
                                           mha.out_proj.weight, mha.out_proj.bias, mha.num_heads, p_attn, p_out,
                                           _fresh_seed() if p_attn > 0.0 else 0, _fresh_seed() if p_out > 0.0 else 0)


# --------------------------------------------------------------------------------------------
# Gate + update + output projection of an edge-attention block as ONE autograd node
# --------------------------------------------------------------------------------------------
class _GatedUpdateTail(torch.autograd.Function):
    """``res + drop(out_proj(agg + sigmoid(lin_ih(agg) + lin_hh(c)) * (lin_self(c) - agg)))``
    (``models/local_encoder.py:183-188,197-199``; ``models/global_interactor.py:126-131,138-140``) from
    the aggregated messages ``agg`` and the normalised centre rows ``c``: four launches forward
    (lin_ih, lin_hh | lin_self as one product, the gate, out_proj with dropout and residual)."""

    @staticmethod
    def forward(ctx, agg, center, res, w_ih, b_ih, w_hh, b_hh, w_self, b_self, w_out, b_out, p_out, seed_out):
        _need_gpu(agg, center, res, w_ih, w_out)
        n, d = agg.shape
        dev = agg.device
        st = _stream(agg)
        agg2, c2, r2 = _rows2(agg), _rows2(center), _rows2(res)
        w_hs = _stacked(w_hh, w_self)
        b_hs = _stacked(b_hh, b_self)
        w_ih_c, w_out_c, b_ih_c, b_out_c = _rows2(w_ih), _rows2(w_out), b_ih.contiguous(), b_out.contiguous()
        ih = torch.empty(n, d, dtype=torch.float32, device=dev)
        hs = torch.empty(n, 2 * d, dtype=torch.float32, device=dev)
        upd = torch.empty(n, d, dtype=torch.float32, device=dev)
        y = torch.empty(n, d, dtype=torch.float32, device=dev)

        def linear(x, x_ld, w, b, rows_out, out, drop=0.0, seed=0, residual=None, res_ld=0):
            _lib.call("hivt_linear_fwd", _p(x), x_ld, _p(w), w.stride(0), _p(b), None, 0, None, 0, None, None, 0.0,
                      EPI_NONE, 1.0, float(drop), int(seed), _seed_ptr(drop), _p(residual), res_ld, n, rows_out, d,
                      _p(out), rows_out, None, 0, None, st)

        linear(agg2, agg2.stride(0), w_ih_c, b_ih_c, d, ih)
        linear(c2, c2.stride(0), w_hs, b_hs, 2 * d, hs)
        _lib.call("hivt_gated_update_fwd", _p(agg2), agg2.stride(0), _p(ih), d, _p(hs), 2 * d, _p(hs, 4 * d), 2 * d, n, d,
                  _p(upd), st)
        linear(upd, d, w_out_c, b_out_c, d, y, p_out, seed_out, r2, r2.stride(0))
        ctx.save_for_backward(agg2, c2, ih, hs, upd, w_ih_c, w_hs, w_out_c, DEVICE_SEED if p_out > 0.0 else None)
        ctx.cfg = (float(p_out), int(seed_out), tuple(res.shape))
        ctx.param_leaves = _all_leaves(w_ih, b_ih, w_hh, b_hh, w_self, b_self, w_out, b_out)
        ctx.res_leaf = res.is_leaf
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        agg, c, ih, hs, upd, w_ih, w_hs, w_out, seed_dev = ctx.saved_tensors
        p_out, seed_out, res_shape = ctx.cfg
        n, d = agg.shape
        dev = agg.device
        st = _stream(agg)
        dy2 = _rows2(dy)
        if p_out > 0.0:
            dz = torch.empty(n, d, dtype=torch.float32, device=dev)
            _lib.call("hivt_dropout_bwd", _p(dy2), dy2.stride(0), None, 0, 0, p_out, seed_out, _p(seed_dev), n, d, _p(dz),
                      d, st)
        else:
            dz = dy2

        def wgrad(g, g_ld, x, x_ld, rows_out):
            return _weight_grad(g, g_ld, x, x_ld, n, rows_out, d, dev, leaves=ctx.param_leaves)

        def data_grad(g, g_ld, w, rows_out, add=None):
            dx = torch.empty(n, d, dtype=torch.float32, device=dev)
            _lib.call("hivt_linear_bwd_data", _p(g), g_ld, _p(w), w.stride(0), n, rows_out, d, None, 0, 1.0, _p(add),
                      d if add is not None else 0, _p(dx), d, st)
            return dx

        dw_out, db_out = wgrad(dz, dz.stride(0), upd, d, d)
        d_upd = data_grad(dz, dz.stride(0), w_out, d)
        d_agg_direct = torch.empty(n, d, dtype=torch.float32, device=dev)
        d_ih = torch.empty(n, d, dtype=torch.float32, device=dev)
        d_hs = torch.empty(n, 2 * d, dtype=torch.float32, device=dev)
        _lib.call("hivt_gated_update_bwd", _p(d_upd), _p(agg), agg.stride(0), _p(ih), d, _p(hs), 2 * d, _p(hs, 4 * d),
                  2 * d, n, d, _p(d_agg_direct), _p(d_ih), _p(d_hs), 2 * d, _p(d_hs, 4 * d), 2 * d, st)
        dw_ih, db_ih = wgrad(d_ih, d, agg, agg.stride(0), d)
        d_agg = data_grad(d_ih, d, w_ih, d, add=d_agg_direct)  # both routes into the aggregated messages
        dw_hs, db_hs = wgrad(d_hs, 2 * d, c, c.stride(0), 2 * d)
        d_c = data_grad(d_hs, 2 * d, w_hs, 2 * d)
        return (d_agg, d_c, _pass_through_grad(dy, res_shape, ctx.res_leaf), dw_ih, db_ih, dw_hs[:d], db_hs[:d], dw_hs[d:], db_hs[d:], dw_out,
                db_out, None, None)


def gated_update_tail(agg: torch.Tensor, center: torch.Tensor, residual: torch.Tensor, lin_ih, lin_hh, lin_self,
                      out_proj, p_out: float, training: bool) -> torch.Tensor:
    """Gate, update, output projection, dropout and residual of an edge-attention block as one autograd
    node (``agg``, ``center``, ``residual`` [N, D] GPU tensors)."""
    p = float(p_out) if training else 0.0
    return _GatedUpdateTail.apply(agg, center, residual, lin_ih.weight, lin_ih.bias, lin_hh.weight, lin_hh.bias,
                                  lin_self.weight, lin_self.bias, out_proj.weight, out_proj.bias, p,
                                  _fresh_seed() if p > 0.0 else 0)


# --------------------------------------------------------------------------------------------
# A whole edge-attention block as ONE autograd node
# --------------------------------------------------------------------------------------------
class GradSink:
    """Running sum of the gradients that several attention blocks produce for ONE tensor of edge rows they all
    read (the interaction module's relative-pose embeddings feed every one of its layers).  Each block's data-gradient
    product adds the sum so far in its store epilogue and leaves its result here instead of returning it; the node
    that handed the rows out (``share_edge_rows``) returns the total.  Autograd would sum the blocks' gradients with
    one elementwise launch per extra reader, each a pass over three tensors of the edge rows' size.
    A sum belongs to the backward pass that formed it (autograd's graph-task id): a pass that never reaches the
    sharing node -- one that asks for other tensors' gradients only -- leaves a sum behind that no later pass
    picks up."""

    def __init__(self) -> None:
        self._total: Optional[torch.Tensor] = None
        self._task = -1

    @property
    def total(self) -> Optional[torch.Tensor]:
        """The sum that the current backward pass has formed so far (None: nothing yet)."""
        return self._total if self._task == torch._C._current_graph_task_id() else None

    @total.setter
    def total(self, value: Optional[torch.Tensor]) -> None:
        self._total, self._task = value, torch._C._current_graph_task_id()

    def take(self) -> Optional[torch.Tensor]:
        total = self.total
        self._total, self._task = None, -1
        return total


class _SharedEdgeRows(torch.autograd.Function):

    @staticmethod
    def forward(ctx, rows, sink):
        ctx.sink = sink
        ctx.set_materialize_grads(False)
        return rows.view_as(rows)

    @staticmethod
    def backward(ctx, grad):
        # runs after every reader of this pass: the blocks' share is in the sink, other readers' arrives here
        total = ctx.sink.take()
        if grad is not None:
            total = grad if total is None else total + grad
        return total, None


def share_edge_rows(rows: torch.Tensor):
    """``(rows, sink)`` for edge rows that several ``attention_block`` calls read: pass ``sink`` to each of them as
    ``edge_grad_sink`` and the rows' gradient is summed inside the blocks' own products (see ``GradSink``)."""
    sink = GradSink()
    return _SharedEdgeRows.apply(rows, sink), sink


class _AttentionBlock(torch.autograd.Function):
    """``res + drop(out_proj(agg + sigmoid(lin_ih(agg) + lin_hh(c)) * (lin_self(c) - agg)))`` with
    ``agg = seg_attn(lin_q(c), [lin_k | lin_v](edge rows) [+ [lin_k_node | lin_v_node](c)[src]], rowptr)``
    (``models/local_encoder.py:166-199, 356-394``; ``models/global_interactor.py:107-140``).

    Every projection of the centre rows ``c`` -- query, (key | value node terms,) hh, self -- is a column block of ONE
    product ``c [w_q; (w_k_node; w_v_node;) w_hh; w_self]^T`` whose blocks the attention and gate kernels read in
    place through their row strides; backward fills ONE gradient ``dA`` of that product block by block (the
    attention kernel writes dq, the per-source sums of dK | dV the node terms, the gate its two blocks) and takes the
    centre rows' gradient and the stacked weights' gradient from it in one product each.  Against separate nodes
    per projection that is two (four) launches fewer per pass and no summation of ``c``'s gradient routes.
    Forward values equal the separate nodes' bit for bit; the centre rows' gradient is one reduction over all
    blocks instead of a sum of rounded partial products."""

    @staticmethod
    def forward(ctx, center, res, edge_rows, rowptr, src, rowptr_by_src, perm_by_src, cfg, *params):
        num_heads, p_attn, seed_attn, p_out, seed_out, tag, padded, node_terms, row_order, ctx.edge_sink = cfg
        if node_terms:
            (w_q, b_q, w_kn, b_kn, w_vn, b_vn, w_hh, b_hh, w_self, b_self, w_k, b_k, w_v, b_v, w_ih, b_ih, w_out,
             b_out) = params
            centre_w, centre_b = [w_q, w_kn, w_vn, w_hh, w_self], [b_q, b_kn, b_vn, b_hh, b_self]
        else:
            w_q, b_q, w_hh, b_hh, w_self, b_self, w_k, b_k, w_v, b_v, w_ih, b_ih, w_out, b_out = params
            centre_w, centre_b = [w_q, w_hh, w_self], [b_q, b_hh, b_self]
        _need_gpu(center, res, edge_rows, w_q, w_out)
        if center.dim() != 2 or edge_rows.dim() != 2:
            raise ValueError("attention_block: centre rows [N, D] and edge rows [E, K]")
        n, d = center.shape
        dev, st = center.device, _stream(center)
        c2, r2 = _rows2(center), _rows2(res)
        wc, bc = _stacked_all(centre_w), _stacked_all(centre_b)
        nc = wc.shape[0]
        w_ih_c, w_out_c, b_ih_c, b_out_c = _rows2(w_ih), _rows2(w_out), b_ih.contiguous(), b_out.contiguous()

        def linear(x, w, b, rows_out, out, drop=0.0, seed=0, residual=None):
            _lib.call("hivt_linear_fwd", _p(x), _ld(x), _p(w), w.stride(0), _p(b), None, 0, None, 0, None, None, 0.0,
                      EPI_NONE, 1.0, float(drop), int(seed), _seed_ptr(drop), _p(residual),
                      _ld(residual) if residual is not None else 0, n, rows_out, d, _p(out), rows_out, None, 0, None, st)

        a = torch.empty(n, nc, dtype=torch.float32, device=dev)
        linear(c2, wc, bc, nc, a)
        q, hs = a[:, :d], a[:, nc - 2 * d:]
        proj, attn = _StandInContext(), _StandInContext()
        if node_terms:
            kv = _PairLinear.forward(proj, edge_rows, w_k, b_k, w_v, b_v, a[:, d:3 * d], src, rowptr_by_src, perm_by_src)
        else:
            kv = _PairLinear.forward(proj, edge_rows, w_k, b_k, w_v, b_v)
        agg = _SegAttn.forward(attn, q, kv, None, rowptr, None, None, None, num_heads, p_attn, seed_attn, tag, padded,
                               False, row_order)
        ih = torch.empty(n, d, dtype=torch.float32, device=dev)
        upd = torch.empty(n, d, dtype=torch.float32, device=dev)
        y = torch.empty(n, d, dtype=torch.float32, device=dev)
        linear(agg, w_ih_c, b_ih_c, d, ih)
        _lib.call("hivt_gated_update_fwd", _p(agg), d, _p(ih), d, _p(hs), _ld(hs), _p(hs, 4 * d), _ld(hs), n, d, _p(upd), st)
        linear(upd, w_out_c, b_out_c, d, y, p_out, seed_out, r2)
        own = (c2, a, ih, upd, wc, w_ih_c, w_out_c, DEVICE_SEED if p_out > 0.0 else None)
        ctx.save_for_backward(*own, *proj.saved_tensors, *attn.saved_tensors)
        ctx.split = (len(own), len(own) + len(proj.saved_tensors))
        ctx.stages = (proj.cfg, attn.cfg)
        ctx.cfg = (float(p_out), int(seed_out), tuple(res.shape), bool(node_terms))
        ctx.param_leaves = _all_leaves(*params)
        ctx.res_leaf = res.is_leaf
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        c2, a, ih, upd, wc, w_ih, w_out, seed_dev = saved[:ctx.split[0]]
        proj_cfg, attn_cfg = ctx.stages
        p_out, seed_out, res_shape, node_terms = ctx.cfg
        need = ctx.needs_input_grad
        n, d = c2.shape
        nc = wc.shape[0]
        dev, st = c2.device, _stream(c2)
        leaves = ctx.param_leaves
        dy2 = _rows2(dy)
        if p_out > 0.0:
            dz = torch.empty(n, d, dtype=torch.float32, device=dev)
            _lib.call("hivt_dropout_bwd", _p(dy2), _ld(dy2), None, 0, 0, p_out, seed_out, _p(seed_dev), n, d, _p(dz), d, st)
        else:
            dz = dy2

        def wgrad(g, x, rows_out):
            return _weight_grad(g, _ld(g), x, _ld(x), n, rows_out, d, dev, leaves=leaves)

        def data_grad(g, w, rows_out, add=None):
            dx = torch.empty(n, d, dtype=torch.float32, device=dev)
            _lib.call("hivt_linear_bwd_data", _p(g), _ld(g), _p(w), w.stride(0), n, rows_out, d, None, 0, 1.0, _p(add),
                      d if add is not None else 0, _p(dx), d, st)
            return dx

        attn = _StandInContext((True, True) + (False,) * 11)
        attn.saved_tensors, attn.cfg = saved[ctx.split[1]:], attn_cfg
        agg = attn.saved_tensors[7]  # the attention's output (see _SegAttn.forward)
        hs = a[:, nc - 2 * d:]
        dw_out, db_out = wgrad(dz, upd, d)
        d_upd = data_grad(dz, w_out, d)
        da = torch.empty(n, nc, dtype=torch.float32, device=dev)
        d_hs = da[:, nc - 2 * d:]
        d_agg_direct = torch.empty(n, d, dtype=torch.float32, device=dev)
        d_ih = torch.empty(n, d, dtype=torch.float32, device=dev)
        _lib.call("hivt_gated_update_bwd", _p(d_upd), _p(agg), d, _p(ih), d, _p(hs), _ld(hs), _p(hs, 4 * d), _ld(hs), n, d,
                  _p(d_agg_direct), _p(d_ih), _p(d_hs), _ld(d_hs), _p(d_hs, 4 * d), _ld(d_hs), st)
        dw_ih, db_ih = wgrad(d_ih, agg, d)
        d_agg = data_grad(d_ih, w_ih, d, add=d_agg_direct)  # both routes into the aggregated messages

        proj = _StandInContext((need[2],) + (True,) * 4 + (node_terms,) + (False,) * 3)
        proj.saved_tensors, proj.cfg, proj.param_leaves = saved[ctx.split[0]:ctx.split[1]], proj_cfg, leaves
        x2, w_kv = proj.saved_tensors[0], proj.saved_tensors[1]
        attn.dq_out = da[:, :d]
        attn.dkv_half = (_half_gradients() and x2.shape[1] % 4 == 0 and w_kv.shape[0] % 8 == 0
                         and x2.stride(0) % 4 == 0 and w_kv.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0
                         and w_kv.data_ptr() % 16 == 0)
        dkv = _SegAttn.backward(attn, d_agg)[1]
        if node_terms:
            proj.d_table_out = da[:, d:3 * d]
        sink = ctx.edge_sink if need[2] else None
        if sink is not None:
            proj.dx_add = sink.total
        d_edge, dw_k, db_k, dw_v, db_v = _PairLinear.backward(proj, dkv)[:5]
        if sink is not None:
            # (the sum so far went into this product's epilogue; the node that shared the rows returns the total)
            sink.total, d_edge = d_edge.reshape(x2.shape), None

        dwc, dbc = wgrad(da, c2, nc)
        d_center = None
        if need[0]:
            d_center = torch.empty(n, d, dtype=torch.float32, device=dev)
            _lib.call("hivt_linear_bwd_data", _p(da), _ld(da), _p(wc), wc.stride(0), n, nc, d, None, 0, 1.0, None, 0,
                      _p(d_center), d, st)
        blocks_w = [dwc[i * d:(i + 1) * d] for i in range(nc // d)]
        blocks_b = [dbc[i * d:(i + 1) * d] for i in range(nc // d)]
        centre = [g for pair in zip(blocks_w, blocks_b) for g in pair]
        return (d_center, _pass_through_grad(dy, res_shape, ctx.res_leaf), d_edge, None, None, None, None, None,
                *centre, dw_k, db_k, dw_v, db_v, dw_ih, db_ih, dw_out, db_out)


def attention_block_supported(center: torch.Tensor, edge_rows: torch.Tensor, residual: Optional[torch.Tensor]) -> bool:
    """Whether ``attention_block`` takes these operands (GPU float32, 2-D, a residual, a width the kernels stack)."""
    return (residual is not None and center.is_cuda and center.dtype == torch.float32 and center.dim() == 2
            and edge_rows.is_cuda and edge_rows.dtype == torch.float32 and edge_rows.dim() == 2
            and center.shape[1] % 4 == 0)


def attention_block(center: torch.Tensor, residual: torch.Tensor, edge_rows: torch.Tensor, rowptr: torch.Tensor,
                    lin_q, lin_k, lin_v, lin_ih, lin_hh, lin_self, out_proj, num_heads: int, p_attn: float, p_out: float,
                    training: bool, tag: str = "", padded: bool = False, lin_k_node=None, lin_v_node=None,
                    src: Optional[torch.Tensor] = None, rowptr_by_src: Optional[torch.Tensor] = None,
                    perm_by_src: Optional[torch.Tensor] = None, row_order: Optional[torch.Tensor] = None,
                    edge_grad_sink: Optional[GradSink] = None) -> torch.Tensor:
    """One edge-attention block -- query / key / value projections, segment-softmax attention over the CSR rows,
    gate, update, output projection, dropout and residual -- as one autograd node.  ``center`` [N, D]: the
    normalised target rows; ``edge_rows`` [E, K]: the per-edge embeddings in CSR order; ``residual`` [N, D].  With
    ``lin_k_node`` / ``lin_v_node`` (the interaction module) an edge's key | value also receive the node terms of its
    source ``src[e]`` (int32, CSR order; the CSR over sources gives their gradient), and sources are the centre rows.
    ``row_order`` (int32 [N], from ``csr_build(..., order=True)``): the order in which the attention launches start
    their rows -- a schedule, the results do not depend on it.  ``edge_grad_sink``: ``edge_rows`` came from
    ``share_edge_rows`` and their gradient is summed across the blocks that read them."""
    pa = float(p_attn) if training else 0.0
    po = float(p_out) if training else 0.0
    node_terms = lin_k_node is not None
    if node_terms and (src is None or src.dtype != torch.int32 or rowptr_by_src is None or perm_by_src is None):
        raise ValueError("attention_block: node terms need int32 source ids and the CSR over sources")
    params = [lin_q.weight, lin_q.bias]
    if node_terms:
        params += [lin_k_node.weight, lin_k_node.bias, lin_v_node.weight, lin_v_node.bias]
    params += [lin_hh.weight, lin_hh.bias, lin_self.weight, lin_self.bias, lin_k.weight, lin_k.bias, lin_v.weight,
               lin_v.bias, lin_ih.weight, lin_ih.bias, out_proj.weight, out_proj.bias]
    cfg = (int(num_heads), pa, _fresh_seed() if pa > 0.0 else 0, po, _fresh_seed() if po > 0.0 else 0, tag,
           bool(padded), node_terms, row_order, edge_grad_sink)
    return _AttentionBlock.apply(center, residual, edge_rows, rowptr, src, rowptr_by_src, perm_by_src, cfg, *params)


# --------------------------------------------------------------------------------------------
# The decoder's two input layers over [global | local] rows without materialising the concatenation
# --------------------------------------------------------------------------------------------
_tile_indices = {}


def _tile_index(modes: int, n: int, agent_major: bool, device):
    """For rows that pair every one of ``modes`` global rows with the local row of their agent: ``idx`` int32
    [modes * n] = the agent of every row, and the grouping of the rows by agent (``rowptr`` [n + 1], ``perm``) that
    the sum over an agent's rows runs in (ascending mode)."""
    key = (modes, n, bool(agent_major), str(device))
    hit = _tile_indices.get(key)
    if hit is None:
        if len(_tile_indices) > 32:
            _tile_indices.clear()
        rows = torch.arange(modes * n, dtype=torch.int64)
        idx = (rows // modes) if agent_major else (rows % n)
        perm = rows if agent_major else (rows % modes) * n + rows // modes  # grouped position a * modes + f -> row
        rowptr = torch.arange(n + 1, dtype=torch.int64) * modes
        hit = _tile_indices[key] = tuple(t.to(torch.int32).to(device) for t in (idx, rowptr, perm))
    return hit


class _DecoderInputs(torch.autograd.Function):
    """``(LN_ReLU(cat(g, l_tiled) W_a^T + b_a), LN_ReLU(cat(l_tiled, g) W_p^T + b_p))`` (``models/decoder.py:117-126``:
    ``aggr_embed`` and the first layer of ``pi``) for g [modes * N, D] global rows and l [N, D] local rows, each
    local row paired with the ``modes`` global rows of its agent.  A Linear over a concatenation is the sum of two
    products, and the local one is the same for all modes of an agent: it runs once per agent ([N, D] x [D, D]) and
    enters the global product's epilogue as an added row -- neither concatenation exists, and the big product
    reduces over D instead of 2 D."""

    @staticmethod
    def forward(ctx, g, l, modes, agent_major, w_a, b_a, gamma_a, beta_a, w_p, b_p, gamma_p, beta_p, eps):
        _need_gpu(g, l, w_a, w_p)
        g2, l2 = _rows2(g), _rows2(l)
        m, d = g2.shape
        n = l2.shape[0]
        if m != modes * n or w_a.shape != (d, 2 * d) or w_p.shape != (d, 2 * d) or l2.shape[1] != d:
            raise ValueError("decoder inputs: expected g [modes * N, D], l [N, D] and two [D, 2 D] weights")
        dev = g.device
        st = _stream(g)
        idx, _, _ = _tile_index(modes, n, agent_major, dev)
        w_a_c, w_p_c = w_a.contiguous(), w_p.contiguous()
        params = [t.contiguous() for t in (b_a, gamma_a, beta_a, b_p, gamma_p, beta_p)]
        b_a_c, gamma_a_c, beta_a_c, b_p_c, gamma_p_c, beta_p_c = params
        fb = 4 * d  # byte offset of a weight's second column half
        outs = []
        # (weight half that multiplies g, half that multiplies l): aggr_embed reads [g | l], pi reads [l | g]
        for w, off_g, off_l, bias, gamma, beta in ((w_a_c, 0, fb, b_a_c, gamma_a_c, beta_a_c),
                                                   (w_p_c, fb, 0, b_p_c, gamma_p_c, beta_p_c)):
            t = torch.empty(n, d, dtype=torch.float32, device=dev)
            _lib.call("hivt_linear_fwd", _p(l2), l2.stride(0), _p(w, off_l), 2 * d, _p(bias), None, 0, None, 0, None, None,
                      0.0, EPI_NONE, 1.0, 0.0, 0, None, None, 0, n, d, d, _p(t), d, None, 0, None, st)
            y = torch.empty(m, d, dtype=torch.float32, device=dev)
            z = torch.empty(m, d, dtype=torch.float32, device=dev)
            stats = torch.empty(m, 2, dtype=torch.float32, device=dev)
            _lib.call("hivt_linear_fwd", _p(g2), g2.stride(0), _p(w, off_g), 2 * d, None, _p(t), d, _p(idx), n,
                      _p(gamma), _p(beta), float(eps), EPI_LN_RELU, 1.0, 0.0, 0, None, None, 0, m, d, d, _p(y), d, _p(z), d,
                      _p(stats), st)
            outs += [y, z, stats]
        ctx.save_for_backward(g2, l2, w_a_c, w_p_c, gamma_a_c, beta_a_c, gamma_p_c, beta_p_c, outs[1], outs[2], outs[4],
                              outs[5])
        ctx.cfg = (int(modes), bool(agent_major), tuple(g.shape), tuple(l.shape))
        ctx.leaves = (_all_leaves(w_a, b_a), _all_leaves(w_p, b_p))
        ctx.norm_leaves = (_all_leaves(gamma_a, beta_a), _all_leaves(gamma_p, beta_p))
        ctx.set_materialize_grads(False)
        return outs[0], outs[3]

    @staticmethod
    @once_differentiable
    def backward(ctx, dy_a, dy_p):
        g2, l2, w_a, w_p, gamma_a, beta_a, gamma_p, beta_p, z_a, stats_a, z_p, stats_p = ctx.saved_tensors
        modes, agent_major, g_shape, l_shape = ctx.cfg
        m, d = g2.shape
        n = l2.shape[0]
        dev = g2.device
        st = _stream(g2)
        _, rowptr, perm = _tile_index(modes, n, agent_major, dev)
        fb = 4 * d
        dg = dl = None
        grads = {}
        items, keep, checks = [], [], []
        for tag, dy, w, off_g, off_l, gamma, beta, z, stats, leaves, norm_leaves in (
                ("a", dy_a, w_a, 0, fb, gamma_a, beta_a, z_a, stats_a, ctx.leaves[0], ctx.norm_leaves[0]),
                ("p", dy_p, w_p, fb, 0, gamma_p, beta_p, z_p, stats_p, ctx.leaves[1], ctx.norm_leaves[1])):
            if dy is None:
                grads[tag] = (None, None, None, None)
                continue
            dy2 = _rows2(dy)
            dz = torch.empty(m, d, dtype=torch.float32, device=dev)
            dgamma, dbeta = _ln_backward(dy2, z, d, stats, gamma, beta, 1, m, d, None, dz, st, norm_leaves)
            # the local product's gradient rows: every agent's sum over its modes, in a fixed order
            dt = torch.empty(n, d, dtype=torch.float32, device=dev)
            _lib.call("hivt_segment_sum_rows", _p(dz), d, d, _p(rowptr), _p(perm), n, _p(dt), d, st)
            if ctx.needs_input_grad[0]:
                nxt = torch.empty(m, d, dtype=torch.float32, device=dev)
                _lib.call("hivt_linear_bwd_data", _p(dz), d, _p(w, off_g), 2 * d, m, d, d, None, 0, 1.0, _p(dg),
                          d if dg is not None else 0, _p(nxt), d, st)
                dg = nxt
            if ctx.needs_input_grad[1]:
                nxt = torch.empty(n, d, dtype=torch.float32, device=dev)
                _lib.call("hivt_linear_bwd_data", _p(dt), d, _p(w, off_l), 2 * d, n, d, d, None, 0, 1.0, _p(dl),
                          d if dl is not None else 0, _p(nxt), d, st)
                dl = nxt
            # the two column halves of the [D, 2 D] weight gradient are two products; the bias gradient is the
            # column sum of dz
            dw = torch.empty(d, 2 * d, dtype=torch.float32, device=dev)
            db = torch.empty(d, dtype=torch.float32, device=dev)
            pair = [[dz.data_ptr(), d, g2.data_ptr(), g2.stride(0), m, d, d, dw.data_ptr() + off_g, db.data_ptr(), 16],
                    [dt.data_ptr(), d, l2.data_ptr(), l2.stride(0), n, d, d, dw.data_ptr() + off_l, 0, 16]]
            if _may_defer(leaves):
                _schedule_flush()
                _deferred.append((pair[0], dev, (dz, g2), ((dw.data_ptr(), db.data_ptr()), leaves)))
                _deferred.append((pair[1], dev, (dt, l2), None))
            else:
                items += pair[0] + pair[1]
                keep += [dz, dt]
            grads[tag] = (dw, db, dgamma, dbeta)
        if items:
            _launch_wgrad_group(items, len(items) // 10, dev)
        ga, gp = grads["a"], grads["p"]
        return (dg.view(g_shape) if dg is not None else None, dl.view(l_shape) if dl is not None else None, None, None,
                ga[0], ga[1], ga[2], ga[3], gp[0], gp[1], gp[2], gp[3], None)


def decoder_inputs(g: torch.Tensor, l: torch.Tensor, modes: int, agent_major: bool, aggr_lin: torch.nn.Linear,
                   aggr_norm: torch.nn.LayerNorm, pi_lin: torch.nn.Linear, pi_norm: torch.nn.LayerNorm
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(relu(aggr_norm(aggr_lin(cat(g, l_tiled)))), relu(pi_norm(pi_lin(cat(l_tiled, g)))))`` for global rows g
    [modes * N, D] and local rows l [N, D]; ``agent_major``: row ``a * modes + f`` of g belongs to agent a (else row
    ``f * N + a``).  One autograd node, four launches forward."""
    return _DecoderInputs.apply(g, l, int(modes), bool(agent_major), aggr_lin.weight, aggr_lin.bias, aggr_norm.weight,
                                aggr_norm.bias, pi_lin.weight, pi_lin.bias, pi_norm.weight, pi_norm.bias,
                                float(aggr_norm.eps))
