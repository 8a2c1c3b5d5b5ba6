"""Data-parallel driver of the G + D + R1 step: one process per GPU, one flat gradient buffer per
optimizer group, one all-reduce per backward.

The reference wraps the whole LightningModule in DDP with ``find_unused_parameters=True``
(``train.py:69-78``), so every 25 MB bucket of all 2.9 M parameters is reduced after BOTH
backwards of a GAN step.  Here the critic gradients (1.5 MB) are reduced after the critic
backward and the generator gradients (10.2 MB at embed_dim 128) after the generator backward; the
critic gradients produced by the generator backward are discarded by the reference anyway
(``models/hivt.py:312``).  The payloads are latency-bound on xGMI, so each group is packed by ONE
concatenation into a contiguous fp32 buffer, reduced with a single RCCL all-reduce (backend name
"nccl" on ROCm), and handed back to the parameters as views of that buffer.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from .hivt import HiVT


class FlatGradBucket:
    """Gradients of a parameter group packed into one contiguous buffer for one all-reduce.

    After a backward the group's gradients are gathered with a single concatenation (parameters
    without a gradient contribute zeros), reduced, and handed back to the parameters as views of the
    reduced buffer, so the optimizer reads the averaged values without further copies."""

    def __init__(self, params: Iterable[torch.nn.Parameter]) -> None:
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        self.flat: Optional[torch.Tensor] = None
        self._zeros = {}

    def _zero_like(self, p: torch.nn.Parameter) -> torch.Tensor:
        key = (p.numel(), p.device, p.dtype)
        if key not in self._zeros:
            self._zeros[key] = torch.zeros(p.numel(), dtype=p.dtype, device=p.device)
        return self._zeros[key]

    def pack(self) -> torch.Tensor:
        self.flat = torch.cat([p.grad.reshape(-1) if p.grad is not None else self._zero_like(p) for p in self.params])
        return self.flat

    def unpack(self) -> None:
        off = 0
        for p, n in zip(self.params, self.sizes):
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def all_reduce_mean(self, group=None) -> None:
        self.pack()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(dist.get_world_size(group))
        self.unpack()


class GanTrainer:
    """Owns the optimizers, LR schedule and gradient buckets of one rank."""

    def __init__(self, model: HiVT, distributed: Optional[bool] = None) -> None:
        self.model = model
        self.distributed = (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) \
            if distributed is None else distributed
        opts, scheds = model.configure_optimizers()
        self.opt_g = opts[0]
        self.opt_d = opts[1] if len(opts) > 1 else None
        self.schedulers = scheds
        self.buckets: Dict[str, FlatGradBucket] = {}
        if self.distributed:
            self.broadcast_state()
            gen = model.generator_parameters() if model.use_gan else list(model.parameters())
            self.buckets["generator"] = FlatGradBucket(gen)
            if model.use_gan:
                self.buckets["critic"] = FlatGradBucket(model.critic_parameters())

    def broadcast_state(self) -> None:
        """Rank 0's parameters and buffers everywhere (what DDP does at construction)."""
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            dist.broadcast(t.data, src=0)

    def _sync(self, group: str) -> None:
        self.buckets[group].all_reduce_mean()

    def step(self, data) -> Dict[str, torch.Tensor]:
        return self.model.training_step(data, self.opt_g, self.opt_d, self._sync if self.distributed else None)

    def end_epoch(self) -> None:
        for s in self.schedulers:
            s.step()
