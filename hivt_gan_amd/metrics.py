"""Validation metrics with the reference's class names and ``update`` / ``compute`` / ``reset`` protocol
(``metrics/ade.py``, ``fde.py``, ``mr.py``, ``jerk.py``, ``speed_violation.py``, ``diversity.py``), without the
torchmetrics dependency: every metric is a running (sum, count) pair kept on the device of its inputs --
``update`` never reads a value back -- and ``sync`` adds the pairs of all data-parallel ranks, which is
what the reference's ``dist_reduce_fx="sum"`` states do at the end of an epoch."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist


class _MeanMetric:
    """total / count over everything seen since the last ``reset``."""

    def __init__(self) -> None:
        self.total: Optional[torch.Tensor] = None
        self.count: Optional[torch.Tensor] = None

    def reset(self) -> None:
        self.total = self.count = None

    def _add(self, total: torch.Tensor, count) -> None:
        total = total.detach().to(torch.float64)
        count = torch.as_tensor(count, dtype=torch.float64, device=total.device)
        if self.total is None:
            self.total, self.count = total.clone(), count.clone()
        else:
            self.total += total
            self.count += count

    def _pair(self, device) -> torch.Tensor:
        """(total, count) as a float64 pair on ``device``; zeros for a metric that has seen nothing."""
        if self.total is None:
            return torch.zeros(2, dtype=torch.float64, device=device)
        return torch.stack([self.total, self.count]).to(device)

    def sync(self, group=None, device=None) -> None:
        """Adds the states of all ranks (no-op outside an initialised process group).  A rank whose
        metric has seen nothing -- no validation batch, or only batches the metric ignores -- still enters
        the collective, with zeros."""
        if dist.is_available() and dist.is_initialized():
            if device is None:
                device = self.total.device if self.total is not None else (
                    torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu")
            pair = self._pair(device)
            dist.all_reduce(pair, group=group)
            self.total, self.count = pair[0], pair[1]

    def compute(self) -> torch.Tensor:
        if self.total is None:
            return torch.tensor(float("nan"))
        # no sample anywhere: NaN (0 / 0), never an exception
        return (self.total / self.count).to(torch.float32)


class ADE(_MeanMetric):
    """Mean over samples of the time-averaged displacement error; pred, target [B, T, 2] (``metrics/ade.py:11-14``)."""

    def update(self, pred: torch.Tensor, target: torch.Tensor) -> None:
        self._add(torch.norm(pred - target, p=2, dim=-1).mean(dim=-1).sum(), pred.size(0))


class FDE(_MeanMetric):
    """Mean displacement error of the last step (``metrics/fde.py:11-13``)."""

    def update(self, pred: torch.Tensor, target: torch.Tensor) -> None:
        self._add(torch.norm(pred[:, -1] - target[:, -1], p=2, dim=-1).sum(), pred.size(0))


class MR(_MeanMetric):
    """Fraction of samples whose last-step error exceeds ``miss_threshold`` metres (``metrics/mr.py:6-15``)."""

    def __init__(self, miss_threshold: float = 2.0) -> None:
        super().__init__()
        self.miss_threshold = miss_threshold

    def update(self, pred: torch.Tensor, target: torch.Tensor) -> None:
        miss = torch.norm(pred[:, -1] - target[:, -1], p=2, dim=-1) > self.miss_threshold
        self._add(miss.sum(), pred.size(0))


class Jerk(_MeanMetric):
    """Mean squared third difference of the positions; traj [B, T, 2], ignored when T < 4 (``metrics/jerk.py:16-28``)."""

    def update(self, traj: torch.Tensor) -> None:
        if traj.size(1) < 4:
            return
        third = traj[:, 3:] - 3.0 * traj[:, 2:-1] + 3.0 * traj[:, 1:-2] - traj[:, :-3]
        self._add(third.pow(2).sum(dim=-1).mean(dim=-1).sum(), traj.size(0))


class SpeedViolation(_MeanMetric):
    """Fraction of trajectories with a step longer than ``speed_limit`` (``metrics/speed_violation.py:11-27``)."""

    def __init__(self, speed_limit: float = 15.0) -> None:
        super().__init__()
        self.speed_limit = speed_limit

    def update(self, traj: torch.Tensor) -> None:
        speed = torch.norm(traj[:, 1:] - traj[:, :-1], dim=-1)
        self._add((speed > self.speed_limit).any(dim=1).sum(), traj.size(0))


class EndpointDiversity(_MeanMetric):
    """Mean over mode pairs of the batch-mean endpoint distance; trajs [F, B, T, 2], one count per call,
    ignored for a single mode (``metrics/diversity.py:16-36``).  All pairs in one distance matrix."""

    def update(self, trajs: torch.Tensor) -> None:
        modes = trajs.size(0)
        if modes < 2:
            return
        end = trajs[:, :, -1]                                              # [F, B, 2]
        pair = torch.norm(end.unsqueeze(0) - end.unsqueeze(1), dim=-1)    # [F, F, B]
        upper = torch.triu(torch.ones(modes, modes, dtype=torch.bool, device=trajs.device), diagonal=1)
        self._add(pair[upper].mean(dim=-1).sum() / (modes * (modes - 1) // 2), 1)


class ValidationMetrics:
    """The metric set of the reference's ``validation_step`` (``models/hivt.py:347-393``): regression loss
    over all valid agents, minADE / minFDE / minMR of each scene's target agent with the mode chosen by
    final displacement, and the realism metrics of that trajectory."""

    def __init__(self) -> None:
        self.val_reg_loss = _MeanMetric()
        self.minADE, self.minFDE, self.minMR = ADE(), FDE(), MR()
        self.jerk, self.speed_violation, self.endpoint_diversity = Jerk(), SpeedViolation(), EndpointDiversity()

    def _all(self) -> Dict[str, _MeanMetric]:
        return {"val_reg_loss": self.val_reg_loss, "val_minADE": self.minADE, "val_minFDE": self.minFDE,
                "val_minMR": self.minMR, "val_jerk": self.jerk, "val_speed_violation": self.speed_violation,
                "val_endpoint_diversity": self.endpoint_diversity}

    def reset(self) -> None:
        for m in self._all().values():
            m.reset()

    def sync(self, group=None, device=None) -> None:
        """ONE collective for all seven metrics: every rank stacks its (total, count) pairs -- zeros where a
        metric has seen nothing on this rank -- so the ranks cannot fall out of step with each other."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        metrics = list(self._all().values())
        if device is None:
            seen = [m.total.device for m in metrics if m.total is not None]
            device = seen[0] if seen else (torch.device("cuda", torch.cuda.current_device())
                                           if dist.get_backend(group) == "nccl" else torch.device("cpu"))
        packed = torch.stack([m._pair(device) for m in metrics])
        dist.all_reduce(packed, group=group)
        for m, pair in zip(metrics, packed):
            m.total, m.count = pair[0].clone(), pair[1].clone()

    def compute(self) -> Dict[str, torch.Tensor]:
        return {k: m.compute() for k, m in self._all().items()}
