"""The step's objectives in single launches: the fused supervised loss (``laplace.hip``) and the adversarial terms
of all critic scales (``critic.hip``)."""
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
# Fused supervised loss
# --------------------------------------------------------------------------------------------
class _SupervisedLoss(torch.autograd.Function):

    @staticmethod
    def forward(ctx, y_hat, pi, y, padding_mask, t_hist, want_best_traj):
        _need_gpu(y_hat, pi, y, padding_mask)
        f, n, hf, ch = y_hat.shape
        if ch != 4 or pi.shape != (n, f) or y.shape != (n, hf, 2):
            raise ValueError("expected y_hat [F,N,H,4], pi [N,F], y [N,H,2]")
        y_hat, pi, y = y_hat.contiguous(), pi.contiguous(), y.contiguous()
        pm = padding_mask.contiguous()
        pm = pm.view(torch.uint8) if pm.dtype == torch.bool else pm
        dev = y_hat.device
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        stats = torch.empty(4, dtype=torch.float32, device=dev)
        best = torch.empty(n, dtype=torch.int64, device=dev)
        soft = torch.empty(n, f, dtype=torch.float32, device=dev)
        ws = _workspace(_lib.query("hivt_laplace_workspace_bytes", n), dev)
        _lib.call("hivt_laplace_minmode_fwd", _p(y_hat), _p(pi), _p(y), _p(pm), pm.stride(0), int(t_hist), f, n, hf,
                  _p(losses), _p(best), _p(soft), _p(stats), _p(ws), ws.numel(), _stream(y_hat))
        traj = None
        if want_best_traj:  # the best mode's locations (models/hivt.py:221-224), cut out next to the argmin
            traj = torch.empty(n, hf, 2, dtype=torch.float32, device=dev)
            _lib.call("hivt_select_mode_fwd", _p(y_hat), _p(best), f, n, hf, 4, _p(traj), _stream(y_hat))
        ctx.save_for_backward(y_hat, pi, y, pm, best, soft, stats)
        ctx.t_hist = int(t_hist)
        ctx.mark_non_differentiable(best)
        ctx.set_materialize_grads(False)
        return losses, best, traj

    @staticmethod
    @once_differentiable
    def backward(ctx, g_losses, _g_best, g_traj):
        y_hat, pi, y, pm, best, soft, stats = ctx.saved_tensors
        f, n, hf, _ = y_hat.shape
        g = g_losses.contiguous().float() if g_losses is not None else \
            torch.zeros(2, dtype=torch.float32, device=y_hat.device)
        d_y_hat = torch.empty_like(y_hat)
        d_pi = torch.empty_like(pi)
        _lib.call("hivt_laplace_minmode_bwd", _p(g), _p(y_hat), _p(pi), _p(y), _p(pm), pm.stride(0), ctx.t_hist, f, n,
                  hf, _p(best), _p(soft), _p(stats), _p(d_y_hat), _p(d_pi), _stream(y_hat))
        if g_traj is not None:  # added where the best mode's locations lie
            g_traj = g_traj.contiguous()
            _lib.call("hivt_select_mode_bwd", _p(g_traj), _p(best), f, n, hf, 4, _p(d_y_hat), _stream(y_hat))
        return d_y_hat, d_pi, None, None, None, None


def supervised_loss(y_hat: torch.Tensor, pi: torch.Tensor, y: torch.Tensor, padding_mask: torch.Tensor,
                    historical_steps: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(reg_loss, cls_loss, best_mode) of the best-of-F Laplace NLL + soft-target CE, without host sync."""
    losses, best, _ = _SupervisedLoss.apply(y_hat, pi, y, padding_mask, historical_steps, False)
    return losses[0], losses[1], best


def supervised_loss_packed(y_hat: torch.Tensor, pi: torch.Tensor, y: torch.Tensor, padding_mask: torch.Tensor,
                           historical_steps: int, want_best_traj: bool = True
                           ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """``(losses [2] = (reg_loss, cls_loss), best_mode [N], best_traj [N, H, 2] or None)``: the same losses as one
    tensor (consumers that take it whole save the two select nodes and their gradient sums) plus the best mode's
    predicted locations ``y_hat[best_mode, arange(N), :, :2]`` (``models/hivt.py:221-224``), whose gradient is added
    inside this node's backward instead of through an index node."""
    return _SupervisedLoss.apply(y_hat, pi, y, padding_mask, historical_steps, bool(want_best_traj))



# --------------------------------------------------------------------------------------------
# Per-scale adversarial loss terms (one launch instead of a dozen framework reductions)
# --------------------------------------------------------------------------------------------
class _HingeR1Loss(torch.autograd.Function):
    """(loss, r1) of one critic scale; gradients flow into the two score vectors and the gradient norms."""

    @staticmethod
    def forward(ctx, s_real, grad_norm2, s_fake, w, count, lambda_r1):
        _need_gpu(s_real, grad_norm2, s_fake, w, count)
        b = s_real.shape[0]
        s_real, grad_norm2, s_fake, w = (t.contiguous() for t in (s_real, grad_norm2, s_fake, w))
        out = torch.empty(2, dtype=torch.float32, device=s_real.device)
        _lib.call("hivt_hinge_r1_fwd", _p(s_real), _p(grad_norm2), _p(s_fake), _p(w), _p(count), b, float(lambda_r1),
                  _p(out), _stream(s_real))
        ctx.save_for_backward(s_real, s_fake, w, count)
        ctx.lambda_r1 = float(lambda_r1)
        loss, r1 = out[0], out[1]
        ctx.mark_non_differentiable(r1)
        return loss, r1

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loss, _g_r1):
        s_real, s_fake, w, count = ctx.saved_tensors
        b = s_real.shape[0]
        g_loss = g_loss.contiguous()
        g_real, g_norm2, g_fake = (torch.empty_like(s_real) for _ in range(3))
        _lib.call("hivt_hinge_r1_bwd", _p(g_loss), _p(s_real), _p(s_fake), _p(w), _p(count), b, ctx.lambda_r1, _p(g_real),
                  _p(g_norm2), _p(g_fake), _stream(s_real))
        return g_real, g_norm2, g_fake, None, None, None


def hinge_r1_loss(s_real: torch.Tensor, grad_norm2: torch.Tensor, s_fake: torch.Tensor, w: torch.Tensor,
                  count: torch.Tensor, lambda_r1: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """``(mean_w relu(1 - s_real) + mean_w relu(1 + s_fake) + lambda_r1 * r1, r1)`` with
    ``r1 = mean_w grad_norm2`` and ``mean_w(v) = sum(w * v) / count`` (``count``: device scalar)."""
    return _HingeR1Loss.apply(s_real, grad_norm2, s_fake, w, count, lambda_r1)


class _WeightedMean(torch.autograd.Function):

    @staticmethod
    def forward(ctx, s, w, count, scale):
        _need_gpu(s, w, count)
        s, w = s.contiguous(), w.contiguous()
        out = torch.empty((), dtype=torch.float32, device=s.device)
        _lib.call("hivt_weighted_mean_fwd", _p(s), _p(w), _p(count), s.shape[0], float(scale), _p(out), _stream(s))
        ctx.save_for_backward(w, count)
        ctx.scale = float(scale)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        w, count = ctx.saved_tensors
        g_s = torch.empty_like(w)
        _lib.call("hivt_weighted_mean_bwd", _p(g_out.contiguous()), _p(w), _p(count), w.shape[0], ctx.scale, _p(g_s),
                  _stream(w))
        return g_s, None, None, None


def weighted_mean(s: torch.Tensor, w: torch.Tensor, count: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """``scale * sum(w * s) / count`` as a 0-d tensor (``count``: device scalar)."""
    return _WeightedMean.apply(s, w, count, scale)



# --------------------------------------------------------------------------------------------
# The adversarial objectives of all critic scales in single launches
# --------------------------------------------------------------------------------------------
class _HingeR1Multi(torch.autograd.Function):
    """``out[0] = sum_k loss_k``, ``out[1 + 2k] = loss_k``, ``out[2 + 2k] = r1_k`` over K critic scales (inputs:
    s_real_0, grad_norm2_0, s_fake_0, s_real_1, ...)."""

    @staticmethod
    def forward(ctx, w, count, lambda_r1, *rows):
        scales = len(rows) // 3
        rows = [t.contiguous() for t in rows]
        _need_gpu(w, *rows)
        b = rows[0].shape[0]
        for t in rows:
            if t.shape != (b,) or t.dtype != torch.float32:
                raise ValueError("hinge_r1_multi: every score / gradient-norm vector must be float32 [B]")
        w = w.contiguous()
        out = torch.empty(2 + 2 * scales, dtype=torch.float32, device=w.device)
        _lib.call("hivt_hinge_r1_multi_fwd", _ptr_array(rows[0::3]), _ptr_array(rows[1::3]), _ptr_array(rows[2::3]), scales,
                  _p(w), _p(count), b, float(lambda_r1), _p(out), _stream(w))
        # (the last element of `out` holds the number of kept rows, given or computed by the kernel)
        ctx.save_for_backward(w, out, *rows[0::3], *rows[2::3])
        ctx.cfg = (scales, float(lambda_r1))
        total, logs = out[0], out[1:1 + 2 * scales]
        ctx.mark_non_differentiable(logs)  # only the total is an objective
        ctx.set_materialize_grads(False)
        return total, logs

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, _g_logs):
        scales, lambda_r1 = ctx.cfg
        w, out, *rest = ctx.saved_tensors
        s_real, s_fake = rest[:scales], rest[scales:]
        b = w.shape[0]
        if g_total is None:
            return (None,) * (3 + 3 * scales)
        g_total = g_total.contiguous()
        count = out[1 + 2 * scales:]
        grads = torch.empty(3 * scales, b, dtype=torch.float32, device=w.device)
        g_real = [grads[3 * k] for k in range(scales)]
        g_norm2 = [grads[3 * k + 1] for k in range(scales)]
        g_fake = [grads[3 * k + 2] for k in range(scales)]
        _lib.call("hivt_hinge_r1_multi_bwd", _p(g_total), _ptr_array(list(s_real)), _ptr_array(list(s_fake)), scales, _p(w),
                  _p(count), b, lambda_r1, _ptr_array(g_real), _ptr_array(g_norm2), _ptr_array(g_fake), _stream(w))
        return (None, None, None, *[grads[i] for i in range(3 * scales)])


def hinge_r1_multi(terms, w: torch.Tensor, count: Optional[torch.Tensor], lambda_r1: float
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """``terms``: per critic scale ``(s_real, grad_norm2, s_fake)`` (at most four scales).  Returns ``(total,
    logs)``: the critics' objective -- the sum over the scales, in order, of ``mean_w relu(1 - s_real) + mean_w
    relu(1 + s_fake) + lambda_r1 * r1`` -- and per scale that term and its ``r1 = mean_w grad_norm2``
    (``mean_w(v) = sum(w * v) / count``; ``logs[2k]``, ``logs[2k + 1]``, no gradient).  ``count`` None: the number of
    kept rows ``max(sum(w), 1)`` is computed by the same launch."""
    if not 1 <= len(terms) <= 4:
        raise ValueError("hinge_r1_multi: one to four scales")
    flat = [t for term in terms for t in term]
    return _HingeR1Multi.apply(w, count, float(lambda_r1), *flat)


class _GeneratorTotal(torch.autograd.Function):
    """``out[0] = (losses[0] + losses[1]) + out[1]``, ``out[1] = lambda_adv * sum_k out[2 + k]``, ``out[2 + k] =
    -mean_w scores_k``."""

    @staticmethod
    def forward(ctx, losses, w, count, lambda_adv, *scores):
        scales = len(scores)
        scores = [t.contiguous() for t in scores]
        _need_gpu(losses, w, *scores)
        losses, w = losses.contiguous(), w.contiguous()
        b = w.shape[0]
        out = torch.empty(3 + scales, dtype=torch.float32, device=w.device)
        _lib.call("hivt_generator_total_fwd", _p(losses), _ptr_array(scores), scales, _p(w), _p(count), b,
                  float(lambda_adv), _p(out), _stream(w))
        ctx.save_for_backward(w, out)
        ctx.cfg = (scales, float(lambda_adv))
        total, parts = out[0], out[1:2 + scales]
        ctx.mark_non_differentiable(parts)
        ctx.set_materialize_grads(False)
        return total, parts

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, _g_parts):
        scales, lambda_adv = ctx.cfg
        w, out = ctx.saved_tensors
        b = w.shape[0]
        if g_total is None:
            return (None,) * (4 + scales)
        g_total = g_total.contiguous()
        count = out[2 + scales:]
        g_losses = torch.empty(2, dtype=torch.float32, device=w.device)
        g_s = torch.empty(scales, b, dtype=torch.float32, device=w.device)
        _lib.call("hivt_generator_total_bwd", _p(g_total), scales, _p(w), _p(count), b, lambda_adv, _p(g_losses),
                  _ptr_array([g_s[k] for k in range(scales)]), _stream(w))
        return (g_losses, None, None, None, *[g_s[k] for k in range(scales)])


def generator_total(losses: torch.Tensor, scores, w: torch.Tensor, count: Optional[torch.Tensor], lambda_adv: float
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
    """The generator's objective and its parts in one launch: ``(total, parts)`` with ``total = (reg_loss + cls_loss) +
    g_adv`` for ``losses = (reg_loss, cls_loss)``, ``parts[0] = g_adv = lambda_adv * sum_k g_k`` and ``parts[1 + k] =
    g_k = -mean_w scores_k`` per critic scale (``models/hivt.py:317-324``; the parts carry no gradient).  ``count``
    None: ``max(sum(w), 1)``, computed by the same launch."""
    if not 1 <= len(scores) <= 4:
        raise ValueError("generator_total: one to four scales")
    return _GeneratorTotal.apply(losses, w, count, float(lambda_adv), *scores)


