"""Driver of the G + D + R1 step on one rank: optimizers, LR schedule, data-parallel gradient
averaging, and the choice of the launch schedule by measurement.

Data parallelism.  The reference wraps the whole LightningModule in DDP with
``find_unused_parameters=True`` (``train.py:69-78``), so every 25 MB bucket of all 2.9 M parameters
is reduced after BOTH backwards of a GAN step.  Here the critic gradients (1.5 MB) are reduced after
the critic backward and the generator gradients (10.2 MB at embed_dim 128) after the generator
backward; the critic gradients produced by the generator backward are discarded by the reference
anyway (``models/hivt.py:312``).  The payloads are latency-bound on xGMI, so each group is packed by
ONE concatenation into a contiguous fp32 buffer, reduced with a single RCCL all-reduce (backend name
"nccl" on ROCm), and handed back to the parameters as views of that buffer.

Host synchronisation.  One step is about 1100 kernel launches of a few microseconds each, so the
host, not the GPU, limits an eagerly launched step, and every read-back stalls the launch stream on
top of that.  With ``edge_capacity`` (see ``GanTrainer.calibrate``) the trainer runs the step with
no read-back at all: the filtered edge lists keep a fixed capacity and every kernel takes the live
count from device memory, the NaN / finite guards and the capacity-overflow flag stay on the device
(``HiVT.sync_free``) and skip the update through the fused AdamW ``found_inf`` input.  ``pad_batch``
additionally brings the node / lane / pair tensors to fixed shapes for callers that need them.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist

from .data import BatchSpec, TemporalData
from .hivt import HiVT


def _is_rccl(group=None) -> bool:
    return dist.get_backend(group) == "nccl"


def _all_reduce_sum_via_host(t: torch.Tensor, group=None) -> None:
    """Sum of a device tensor over the ranks of a CPU backend (gloo): staged through host memory."""
    host = t.detach().cpu()
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    t.copy_(host)


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

    def pack(self, bad: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``bad`` (device bool): this rank's contribution is replaced by zeros (selected, not
        multiplied, so non-finite gradients cannot leak into the sum)."""
        self.flat = torch.cat([p.grad.reshape(-1) if p.grad is not None else self._zero_like(p) for p in self.params])
        if bad is not None:
            self.flat = torch.where(bad, torch.zeros((), dtype=self.flat.dtype, device=self.flat.device), self.flat)
        return self.flat

    def unpack(self) -> None:
        off = 0
        for p, n in zip(self.params, self.sizes):
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def all_reduce_mean(self, group=None, bad: Optional[torch.Tensor] = None) -> None:
        self.pack(bad)
        if _is_rccl(group):  # RCCL averages inside the collective
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)
        else:
            if self.flat.is_cuda:
                _all_reduce_sum_via_host(self.flat, group)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))
        self.unpack()


class GanTrainer:
    """Owns the optimizers, LR schedule and gradient buckets of one rank.

    ``edge_capacity`` (three integers: agent-agent over all history steps, lane-actor, last-step
    interaction) switches the step to the read-back-free mode described in the module docstring.  A
    batch whose filtered edge lists do not fit is discarded on the device and counted in
    ``overflowed_steps()``."""

    def __init__(self, model: HiVT, distributed: Optional[bool] = None,
                 edge_capacity: Optional[Tuple[int, int, int]] = None) -> None:
        self.model = model
        self.distributed = (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) \
            if distributed is None else distributed
        self.edge_capacity = tuple(int(c) for c in edge_capacity) if edge_capacity is not None else None
        if self.edge_capacity is not None:
            if not next(model.parameters()).is_cuda:
                raise RuntimeError("the read-back-free step needs the model on the GPU")
            model.sync_free = True
        if next(model.parameters()).is_cuda:
            model.pack_parameters()  # before the optimizers take hold of the parameters
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
        self._overflows: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def calibrate(model: HiVT, batches: Iterable[TemporalData], slack: float = 1.1, multiple: int = 256) -> BatchSpec:
        """A ``BatchSpec`` covering ``batches`` (GPU tensors): tensor capacities from their shapes and
        filtered-edge capacities from building their edge sets once (one read-back per batch)."""
        from .local_encoder import build_edge_plan
        batches = list(batches)
        spec = BatchSpec.covering(batches)
        worst = [0, 0, 0]
        for b in batches:
            plan = build_edge_plan(b, model.historical_steps, model.local_encoder.local_radius)
            for i, es in enumerate((plan.agent_agent, plan.lane_actor, plan.interaction)):
                worst[i] = max(worst[i], es.num_edges)
        spec.edge_capacity = tuple(int(-(-int(w * slack + 1) // multiple) * multiple) for w in worst)
        return spec

    def broadcast_state(self) -> None:
        """Rank 0's parameters and buffers everywhere (what DDP does at construction)."""
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            if t.is_cuda and not _is_rccl():  # CPU backend: through host memory
                host = t.data.cpu()
                dist.broadcast(host, src=0)
                t.data.copy_(host)
            else:
                dist.broadcast(t.data, src=0)

    def _sync(self, group: str, bad: Optional[torch.Tensor] = None) -> None:
        self.buckets[group].all_reduce_mean(bad=bad)

    def step(self, data) -> Dict[str, torch.Tensor]:
        sync = self._sync if self.distributed else None
        if self.edge_capacity is None:
            return self.model.training_step(data, self.opt_g, self.opt_d, sync)
        data["edge_capacity"] = self.edge_capacity
        logs = self.model.training_step(data, self.opt_g, self.opt_d, sync)
        if self._overflows is None:
            self._overflows = torch.zeros((), dtype=torch.int64, device=data["edge_plan"].overflow.device)
        self._overflows.add_(data["edge_plan"].overflow)  # in place: a recorded step keeps counting when replayed
        return logs

    def tune(self, run_step, rounds: int = 2, steps: int = 4) -> Dict[str, bool]:
        """Picks the launch schedule by measurement: ``run_step(i)`` (one training step on some batch) is
        timed with the critics on one stream and with a stream per critic, alternating ``rounds`` times,
        and the faster setting stays.  Which one wins depends on whether the host or the GPU bounds the
        step on the machine at hand.  Under data parallelism every rank adopts the majority's choice."""
        import time
        from . import losses
        if not torch.cuda.is_available():
            return {"critic_streams": losses.CONCURRENT_CRITICS}
        spent = {True: 0.0, False: 0.0}
        i = 0
        for _ in range(rounds):
            for mode in (True, False):
                losses.CONCURRENT_CRITICS = mode
                run_step(i)
                i += 1
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    run_step(i)
                    i += 1
                torch.cuda.synchronize()
                spent[mode] += time.perf_counter() - t0
        vote = torch.tensor([1.0 if spent[True] <= spent[False] else 0.0], device="cuda")
        if self.distributed:
            if _is_rccl():
                dist.all_reduce(vote)
            else:
                _all_reduce_sum_via_host(vote)
            vote /= dist.get_world_size()
        losses.CONCURRENT_CRITICS = bool(vote.item() >= 0.5)
        return {"critic_streams": losses.CONCURRENT_CRITICS}

    def overflowed_steps(self) -> int:
        """Steps discarded because a filtered edge list exceeded its capacity (reads the device)."""
        return int(self._overflows.item()) if self._overflows is not None else 0

    def end_epoch(self) -> None:
        for s in self.schedulers:
            s.step()
