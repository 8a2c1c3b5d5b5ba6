"""Loss modules with the reference's class names and call signatures (``losses/*.py``), plus the
sync-free forms used by the training step: the fused supervised loss kernel and masked hinge / R1
losses that take all agents with a keep-mask instead of boolean-indexed subsets."""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .critics import multi_supported, scores_multi


# The critics are independent small networks (a few thousand rows each), so one of them fills a fraction
# of the GPU.  On the GPU every critic gets a stream of its own: their forward passes overlap, and so do
# their backward passes (autograd replays each operation on the stream of its forward).
CONCURRENT_CRITICS = os.environ.get("HIVT_CONCURRENT_CRITICS", "1") != "0"
_side_streams: Dict[int, List["torch.cuda.Stream"]] = {}


def _streams(device: torch.device, n: int):
    pool = _side_streams.setdefault(device.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def per_critic(critics: dict, inputs: Sequence[torch.Tensor], body: Callable):
    """``[body(name, critic) for name, critic in critics.items()]`` (tuples of tensors).  For GPU
    inputs each critic runs on its own stream, ordered after the current stream's work so far; the
    current stream then waits for all of them.  ``inputs`` are the tensors ``body`` reads."""
    if not (CONCURRENT_CRITICS and len(critics) > 1 and inputs[0].is_cuda):
        return [body(name, critic) for name, critic in critics.items()]
    main = torch.cuda.current_stream(inputs[0].device)
    streams = _streams(inputs[0].device, len(critics))
    outs = []
    ready = main.record_event()
    for s, (name, critic) in zip(streams, critics.items()):
        s.wait_event(ready)
        with torch.cuda.stream(s):
            outs.append(body(name, critic))
    for s, out in zip(streams, outs):
        main.wait_stream(s)
        for t in out:
            t.record_stream(main)  # produced on s, consumed on the current stream
        for t in inputs:
            t.record_stream(s)     # produced on the current stream, read on s
    return outs


def join_critic_streams(device: torch.device) -> None:
    """Makes the current stream wait for everything queued on the critics' streams (after a backward
    pass, before the optimizer or a collective touches the critics' gradients)."""
    main = torch.cuda.current_stream(device)
    for s in _side_streams.get(device.index, []):
        main.wait_stream(s)


def _reduce(x: torch.Tensor, reduction: str) -> torch.Tensor:
    if reduction == "mean":
        return x.mean()
    if reduction == "sum":
        return x.sum()
    if reduction == "none":
        return x
    raise ValueError(f"{reduction} is not a valid value for reduction")


class LaplaceNLLLoss(nn.Module):
    """``losses/laplace_nll_loss.py:4-36``: pred[..., 2:4] is a log-scale clamped to [-10, 10]."""

    def __init__(self, eps: float = 1e-6, reduction: str = "mean") -> None:
        super().__init__()
        self.eps = eps
        self.reduction = reduction

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        loc, log_scale = pred.chunk(2, dim=-1)
        log_scale = log_scale.clamp(min=-10.0, max=10.0)
        nll = log_scale + math.log(2.0) + (target - loc).abs() / (log_scale.exp() + self.eps)
        return _reduce(nll, self.reduction)


class SoftTargetCrossEntropyLoss(nn.Module):
    """``losses/soft_target_cross_entropy_loss.py:6-23``."""

    def __init__(self, reduction: str = "mean") -> None:
        super().__init__()
        self.reduction = reduction

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return _reduce(-(target * F.log_softmax(pred, dim=-1)).sum(dim=-1), self.reduction)


class R1Regularization(nn.Module):
    """``losses/r1_regularization.py:5-36``: mean over samples of ||d D(x).sum() / dx||^2 (double-differentiable)."""

    def forward(self, d_out: torch.Tensor, real_data: torch.Tensor) -> torch.Tensor:
        (grad,) = torch.autograd.grad(outputs=d_out.sum(), inputs=real_data, create_graph=True, retain_graph=True,
                                      only_inputs=True)
        return grad.pow(2).reshape(grad.size(0), -1).sum(dim=1).mean()


class AdversarialDiscriminatorLoss(nn.Module):
    """Multi-scale hinge loss + R1 on real samples (``losses/adv_losses.py:7-58``)."""

    def __init__(self, lambda_r1: float = 1.0) -> None:
        super().__init__()
        self.lambda_r1 = lambda_r1
        self.r1_reg = R1Regularization()

    def forward(self, critics: dict, real_trajs: dict, fake_trajs: dict):
        total, logs = 0.0, {}
        for name, critic in critics.items():
            real = real_trajs[name].requires_grad_(True)
            d_real, d_fake = critic(real), critic(fake_trajs[name].detach())
            r1 = self.r1_reg(d_real, real)
            loss = F.relu(1.0 - d_real).mean() + F.relu(1.0 + d_fake).mean() + self.lambda_r1 * r1
            total = total + loss
            logs[f"d_loss_{name}"] = loss.detach()
            logs[f"r1_{name}"] = r1.detach()
        return total, logs

    def forward_masked(self, critics: dict, real: torch.Tensor, fake: torch.Tensor, keep: Optional[torch.Tensor],
                       weights: Optional[torch.Tensor] = None, count: Optional[torch.Tensor] = None):
        """Same value as ``forward`` on ``real[keep]`` / ``fake[keep]`` sliced to each critic's horizon,
        computed on all rows with ``keep`` as weights (no data-dependent shapes, no host sync).
        Valid because every term is a per-sample mean and the critics act row-wise.  ``weights`` / ``count``: the
        keep mask as floats and the number of kept rows (at least 1), if the caller has them already."""
        w = weights if weights is not None else keep.to(real.dtype)
        real, fake = real.detach(), fake.detach()
        fused = multi_supported(critics.values(), real, fake) and len(critics) <= 4
        # (the fused objective counts the kept rows itself when it is not handed their number)
        cnt = count if (count is not None or fused) else w.sum().clamp(min=1.0)

        def one(name, critic):
            # score and ||d score / d input||^2 of the real samples come from ONE critic call, as in the
            # reference (hinge and R1 share the forward pass, models/hivt.py:244-252)
            d_real, grad_norm2 = critic.scores_and_input_grad_norm(real)
            d_fake = critic(fake)
            if d_real.is_cuda:  # the three masked means and their sum in one launch
                return ops.hinge_r1_loss(d_real, grad_norm2, d_fake, w, cnt, self.lambda_r1)
            r1 = (grad_norm2 * w).sum() / cnt
            loss = (F.relu(1.0 - d_real) * w).sum() / cnt + (F.relu(1.0 + d_fake) * w).sum() / cnt + self.lambda_r1 * r1
            return loss, r1.detach()

        total, logs = 0.0, {}
        if multi_supported(critics.values(), real, fake):
            # every critic on the real and on the generated rows in one library call (their launches are shared)
            members = list(critics.values())
            outs = scores_multi([(c, real, True) for c in members] + [(c, fake, False) for c in members])
            if len(members) <= 4:  # the scales' terms and their sum in one launch
                total, parts = ops.hinge_r1_multi([(outs[i][0], outs[i][1], outs[len(members) + i])
                                                   for i in range(len(members))], w, cnt, self.lambda_r1)
                for i, name in enumerate(critics):
                    logs[f"d_loss_{name}"] = parts[2 * i]
                    logs[f"r1_{name}"] = parts[2 * i + 1]
                return total, logs
            terms = [ops.hinge_r1_loss(outs[i][0], outs[i][1], outs[len(members) + i], w, cnt, self.lambda_r1)
                     for i in range(len(members))]
        else:
            terms = per_critic(critics, (real, fake, w, cnt), one)
        for name, (loss, r1) in zip(critics, terms):
            total = total + loss
            logs[f"d_loss_{name}"] = loss.detach()
            logs[f"r1_{name}"] = r1
        return total, logs


class AdversarialGeneratorLoss(nn.Module):
    """``losses/adv_losses.py:62-99``."""

    def __init__(self, lambda_adv: float = 1.0) -> None:
        super().__init__()
        self.lambda_adv = lambda_adv

    def forward(self, critics: dict, fake_trajs: dict):
        total, logs = 0.0, {}
        for name, critic in critics.items():
            g = -critic(fake_trajs[name]).mean()
            total = total + g
            logs[f"g_loss_{name}"] = g.detach()
        return self.lambda_adv * total, logs

    def total_masked(self, critics: dict, fake: torch.Tensor, supervised: torch.Tensor, weights: torch.Tensor,
                     count: Optional[torch.Tensor] = None):
        """``(g_total, g_adv, logs)`` with ``g_total = (supervised[0] + supervised[1]) + g_adv``
        (``models/hivt.py:317-324``): the critics' passes in one library call and the objective with all of its
        parts in one launch.  ``supervised`` = (regression, classification) loss as one tensor [2]; ``weights`` /
        ``count`` as in ``forward_masked``.  Needs ``multi_supported`` critics, at most four."""
        outs = scores_multi([(c, fake, False) for c in critics.values()])
        total, parts = ops.generator_total(supervised, outs, weights, count, self.lambda_adv)
        logs = {f"g_loss_{name}": parts[1 + i] for i, name in enumerate(critics)}
        return total, parts[0], logs

    def forward_masked(self, critics: dict, fake: torch.Tensor, keep: Optional[torch.Tensor],
                       weights: Optional[torch.Tensor] = None, count: Optional[torch.Tensor] = None):
        w = weights if weights is not None else keep.to(fake.dtype)
        cnt = count if count is not None else w.sum().clamp(min=1.0)
        total, logs = 0.0, {}
        def one(_, critic):
            scores = critic(fake)
            if scores.is_cuda:
                return (ops.weighted_mean(scores, w, cnt, -1.0),)
            return (-(scores * w).sum() / cnt,)

        if multi_supported(critics.values(), fake):
            outs = scores_multi([(c, fake, False) for c in critics.values()])
            terms = [(ops.weighted_mean(s, w, cnt, -1.0),) for s in outs]
        else:
            terms = per_critic(critics, (fake, w, cnt), one)
        for name, (g,) in zip(critics, terms):
            total = total + g
            logs[f"g_loss_{name}"] = g.detach()
        return self.lambda_adv * total, logs


def fused_supervised_losses(y_hat: torch.Tensor, pi: torch.Tensor, y: torch.Tensor, padding_mask: torch.Tensor,
                            historical_steps: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(reg_loss, cls_loss, best_mode) of ``HiVT._supervised_losses`` (``models/hivt.py:174-209``) through the
    fused min-over-modes / Laplace-NLL / soft-target-CE kernel; zeros when no future step is valid."""
    return ops.supervised_loss(y_hat, pi, y, padding_mask, historical_steps)
