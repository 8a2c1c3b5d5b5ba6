"""Data-parallel driver of the G + D + R1 step: one process per GPU, one flat gradient buffer per
optimizer group, one all-reduce per backward.

The reference wraps the whole LightningModule in DDP with ``find_unused_parameters=True``
(``train.py:69-78``), so every 25 MB bucket of all 2.9 M parameters is reduced after BOTH
backwards of a GAN step.  Here the critic gradients (1.5 MB) are reduced after the critic
backward and the generator gradients (10.2 MB at embed_dim 128) after the generator backward; the
critic gradients produced by the generator backward are discarded by the reference anyway
(``models/hivt.py:312``).  The payloads are latency-bound on xGMI, so each group is ONE
contiguous fp32 buffer that the parameters' ``.grad`` tensors alias (no packing copies), reduced
with a single RCCL all-reduce (backend name "nccl" on ROCm).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from .hivt import HiVT


class FlatGradBucket:
    """All ``.grad`` tensors of a parameter group as views into one contiguous buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter]) -> None:
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def rebind(self) -> None:
        """Re-alias after something replaced a ``.grad`` (e.g. ``zero_grad(set_to_none=True)``)."""
        off = 0
        for p in self.params:
            view = self.flat[off:off + p.numel()].view_as(p)
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                if p.grad is not None:
                    view.copy_(p.grad)
                p.grad = view
            off += p.numel()

    def all_reduce_mean(self, group=None) -> None:
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(dist.get_world_size(group))


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
        bucket = self.buckets[group]
        bucket.rebind()
        bucket.all_reduce_mean()

    def step(self, data) -> Dict[str, torch.Tensor]:
        return self.model.training_step(data, self.opt_g, self.opt_d, self._sync if self.distributed else None)

    def end_epoch(self) -> None:
        for s in self.schedulers:
            s.step()
