"""Spectral normalisation of a stack of linear layers (the critics): all layers of a forward call in one launch
(``spectral.hip``)."""
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
# Spectral normalisation of a stack of linear layers (the critics)
# --------------------------------------------------------------------------------------------
def _round4(n: int) -> int:
    return (n + 3) // 4 * 4


def _aligned_parts(buf: torch.Tensor, sizes):
    parts, off = [], 0
    for n in sizes:
        parts.append(buf[off:off + n])
        off += _round4(n)
    return parts


class _SpectralNormalize(torch.autograd.Function):
    """``(w_0 / sigma_0, ..., w_{L-1} / sigma_{L-1})`` for L weight matrices in one launch; in training
    mode one power iteration updates the ``u`` / ``v`` buffers in place first."""

    @staticmethod
    def forward(ctx, state, *weights):
        us, vs, training, eps = state
        _need_gpu(*weights)
        layers = len(weights)
        ws = [w.contiguous() for w in weights]
        dev = ws[0].device
        rows = (ctypes.c_int32 * layers)(*[w.shape[0] for w in ws])
        cols = (ctypes.c_int32 * layers)(*[w.shape[1] for w in ws])
        wn = [torch.empty_like(w) for w in ws]
        # one buffer for every layer's copies of u and v, each 16-byte aligned
        sizes = [w.shape[0] for w in ws] + [w.shape[1] for w in ws]
        used = torch.empty(sum(_round4(n) for n in sizes), dtype=torch.float32, device=dev)
        parts = _aligned_parts(used, sizes)
        u_used, v_used = parts[:layers], parts[layers:]
        sigma = torch.empty(layers, dtype=torch.float32, device=dev)
        _lib.call("hivt_spectral_norm_fwd", layers, _ptr_array(ws), _ptr_array(us), _ptr_array(vs), rows, cols,
                  1 if training else 0, float(eps), _ptr_array(wn), _ptr_array(u_used), _ptr_array(v_used), _p(sigma),
                  _stream(ws[0]))
        ctx.save_for_backward(used, sigma, *ws)
        ctx.sizes = sizes
        return tuple(wn)

    @staticmethod
    @once_differentiable
    def backward(ctx, *d_wn):
        used, sigma, *ws = ctx.saved_tensors
        layers = len(ws)
        parts = _aligned_parts(used, ctx.sizes)
        u_used, v_used = parts[:layers], parts[layers:]
        keep = [i for i, g in enumerate(d_wn) if g is not None]
        grads = [None] * layers
        if keep:
            gs = [d_wn[i].contiguous() for i in keep]
            outs = [torch.empty_like(ws[i]) for i in keep]
            sig = sigma if len(keep) == layers else sigma[keep]
            rows = (ctypes.c_int32 * len(keep))(*[ws[i].shape[0] for i in keep])
            cols = (ctypes.c_int32 * len(keep))(*[ws[i].shape[1] for i in keep])
            _lib.call("hivt_spectral_norm_bwd", len(keep), _ptr_array(gs), _ptr_array([ws[i] for i in keep]),
                      _ptr_array([u_used[i] for i in keep]), _ptr_array([v_used[i] for i in keep]), _p(sig), rows, cols,
                      _ptr_array(outs), _stream(gs[0]))
            for i, o in zip(keep, outs):
                grads[i] = o
        return (None, *grads)


class _SpectralNormalizeRounds(torch.autograd.Function):
    """``rounds`` successive calls of ``_SpectralNormalize`` on the same weights (every round advances ``u`` / ``v`` by
    one power iteration and normalises with ITS sigma) as one node of the graph: outputs round-major, and the
    gradient of a weight comes out as the sum over the rounds, added in one multi-tensor launch instead of one
    accumulation per weight."""

    @staticmethod
    def forward(ctx, state, *weights):
        us, vs, training, eps, rounds = state
        _need_gpu(*weights)
        layers = len(weights)
        ws = [w.contiguous() for w in weights]
        dev = ws[0].device
        rows = (ctypes.c_int32 * layers)(*[w.shape[0] for w in ws])
        cols = (ctypes.c_int32 * layers)(*[w.shape[1] for w in ws])
        sizes = [w.shape[0] for w in ws] + [w.shape[1] for w in ws]
        per_round = sum(_round4(n) for n in sizes)
        used = torch.empty(rounds * per_round, dtype=torch.float32, device=dev)
        sigma = torch.empty(rounds * layers, dtype=torch.float32, device=dev)
        outs = []
        for r in range(rounds):
            wn = [torch.empty_like(w) for w in ws]
            parts = _aligned_parts(used[r * per_round:(r + 1) * per_round], sizes)
            _lib.call("hivt_spectral_norm_fwd", layers, _ptr_array(ws), _ptr_array(us), _ptr_array(vs), rows, cols,
                      1 if training else 0, float(eps), _ptr_array(wn), _ptr_array(parts[:layers]),
                      _ptr_array(parts[layers:]), _p(sigma[r * layers:]), _stream(ws[0]))
            outs += wn
        ctx.save_for_backward(used, sigma, *ws)
        ctx.sizes, ctx.rounds, ctx.per_round = sizes, rounds, per_round
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *d_wn):
        used, sigma, *ws = ctx.saved_tensors
        layers, rounds = len(ws), ctx.rounds
        grads = [None] * layers
        for r in range(rounds):
            d_r = d_wn[r * layers:(r + 1) * layers]
            keep = [i for i, g in enumerate(d_r) if g is not None]
            if not keep:
                continue
            parts = _aligned_parts(used[r * ctx.per_round:(r + 1) * ctx.per_round], ctx.sizes)
            u_used, v_used = parts[:layers], parts[layers:]
            gs = [d_r[i].contiguous() for i in keep]
            outs = [torch.empty_like(ws[i]) for i in keep]
            sig = sigma[r * layers:(r + 1) * layers]
            if len(keep) != layers:
                sig = sig[keep]
            rows = (ctypes.c_int32 * len(keep))(*[ws[i].shape[0] for i in keep])
            cols = (ctypes.c_int32 * len(keep))(*[ws[i].shape[1] for i in keep])
            _lib.call("hivt_spectral_norm_bwd", len(keep), _ptr_array(gs), _ptr_array([ws[i] for i in keep]),
                      _ptr_array([u_used[i] for i in keep]), _ptr_array([v_used[i] for i in keep]), _p(sig), rows, cols,
                      _ptr_array(outs), _stream(gs[0]))
            have = [i for i in keep if grads[i] is not None]
            if have:  # one multi-tensor launch for the sums over the rounds
                torch._foreach_add_([grads[i] for i in have], [o for i, o in zip(keep, outs) if grads[i] is not None])
            for i, o in zip(keep, outs):
                if grads[i] is None:
                    grads[i] = o
        return (None, *grads)


def spectral_normalize_rounds(weights, us, vs, rounds: int, training: bool, eps: float = 1e-12):
    """``rounds`` calls of ``spectral_normalize`` in a row (what ``rounds`` forward calls of a spectrally normalised
    layer do in training mode) as one node of the autograd graph; returns one list of normalised copies per round."""
    if len(weights) > 16 or rounds < 1:
        raise ValueError("at most 16 layers per call, at least one round")
    for w, u, v in zip(weights, us, vs):
        if w.dim() != 2 or u.shape != (w.shape[0],) or v.shape != (w.shape[1],):
            raise ValueError("weights must be [out, in] with u [out] and v [in]")
        if w.dtype != torch.float32 or not u.is_contiguous() or not v.is_contiguous():
            raise TypeError("spectral_normalize needs float32 weights and contiguous u / v")
    _need_gpu(*us, *vs)
    outs = _SpectralNormalizeRounds.apply((list(us), list(vs), bool(training), eps, int(rounds)), *weights)
    n = len(weights)
    return [list(outs[r * n:(r + 1) * n]) for r in range(rounds)]


def spectral_normalize(weights, us, vs, training: bool, eps: float = 1e-12):
    """Spectrally normalised copies of ``weights`` (a list of [out, in] matrices, at most 16).  ``us`` /
    ``vs`` are the persistent power-iteration vectors; in training mode they advance by one iteration
    (in place), exactly as the hook of ``torch.nn.utils.spectral_norm`` does per forward call."""
    if len(weights) > 16:
        raise ValueError("at most 16 layers per call")
    for w, u, v in zip(weights, us, vs):
        if w.dim() != 2 or u.shape != (w.shape[0],) or v.shape != (w.shape[1],):
            raise ValueError("weights must be [out, in] with u [out] and v [in]")
        if w.dtype != torch.float32 or not u.is_contiguous() or not v.is_contiguous():
            raise TypeError("spectral_normalize needs float32 weights and contiguous u / v")
    _need_gpu(*us, *vs)
    return _SpectralNormalize.apply((list(us), list(vs), bool(training), eps), *weights)


