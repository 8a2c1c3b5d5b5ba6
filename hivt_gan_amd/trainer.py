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

import contextlib
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


# every gradient of a flat bucket starts at a multiple of this many elements
_ALIGN = 32


class FlatGradBucket:
    """Gradients of a parameter group in one contiguous buffer, reduced in a few chunks.

    The buffer holds the group's gradients in parameter order, every gradient starting at a multiple of
    128 bytes (the views handed back to the optimizer take its vectorised path; the gaps stay zero), plus
    one trailing element, the share of ranks that contributed (1 - bad, averaged).  ``chunks`` partitions the parameters in the order their
    gradients become ready during backward (decoder first, agent-agent encoder last).  With ``overlap``
    every chunk is packed and handed to an asynchronous all-reduce from a post-accumulate hook as soon
    as its last gradient has arrived, so the collective runs beside the rest of the backward pass;
    ``finish`` packs what is left, waits, and hands the reduced values back to the parameters as views
    of the buffer.  Without ``overlap`` everything happens in ``finish``: one pack, one all-reduce.

    A rank that discarded its batch contributes ZEROS and the sum is still divided by the world size -- the
    reference's behaviour: its skipped rank returns a zero-valued dummy loss precisely so that it enters the same
    collectives (``models/hivt.py:262-265,301-310``), and DDP then averages over all ranks.  With k of W ranks
    contributing the applied gradient is k / W of the contributors' mean; the trailing element of the buffer
    carries that share k / W (``tests/test_dp_gloo.py`` pins both)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], chunks: Optional[List[List[torch.nn.Parameter]]] = None,
                 overlap: bool = False, group=None) -> None:
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        self.group = group
        self.overlap = bool(overlap)
        self.flat: Optional[torch.Tensor] = None
        self._views: List[torch.Tensor] = []
        offsets, off = {}, 0
        for p, n in zip(self.params, self.sizes):
            offsets[id(p)] = off
            off += -(-n // _ALIGN) * _ALIGN
        self.offsets = [offsets[id(p)] for p in self.params]
        self.total = off  # elements of the gradient part (alignment gaps included)
        if chunks is None:
            chunks = [self.params]
        own = {id(p) for p in self.params}
        chunks = [[p for p in c if id(p) in own] for c in chunks]
        chunks = [c for c in chunks if c]
        flat_order = [id(p) for c in chunks for p in c]
        if flat_order != [id(p) for p in self.params]:
            raise ValueError("chunks must list the group's parameters in the group's order")
        self.chunks = []  # (first element, one past the last, parameters)
        starts = [offsets[id(c[0])] for c in chunks] + [self.total]
        for c, a, b in zip(chunks, starts[:-1], starts[1:]):
            self.chunks.append((a, b, c))
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._chunk_of = {id(p): i for i, (_, _, c) in enumerate(self.chunks) for p in c}
        self._pending: List[int] = []
        self._packed: List[bool] = []
        self._next = -1
        self._work: list = []
        self._bad: Optional[torch.Tensor] = None
        self._armed = False
        self._hooks = []
        if self.overlap:
            from . import ops
            ops.trust_grad_hook(self._on_grad)  # (it issues the deferred weight gradients before it reads any)
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------------------------------
    def _buffer(self) -> torch.Tensor:
        if self.flat is None or self.flat.device != self.params[0].device:
            self.flat = torch.zeros(self.total + 1, dtype=self.params[0].dtype, device=self.params[0].device)
            self._views = [self.flat[o:o + n].view_as(p) for p, o, n in zip(self.params, self.offsets, self.sizes)]
        return self.flat

    def begin(self, bad: Optional[torch.Tensor] = None) -> None:
        """Call before the backward pass whose gradients this bucket reduces.  ``bad`` (device bool): this
        rank's contribution is replaced by zeros (selected, not multiplied, so non-finite gradients cannot
        leak into the sum)."""
        self._buffer()
        self._bad = bad
        self._pending = [len(c) for _, _, c in self.chunks]
        self._packed = [False] * len(self.chunks)
        self._next = len(self.chunks) - 1
        self._work = []
        self._armed = True

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if not (self._armed and self.overlap):
            return
        self._pending[self._chunk_of[id(p)]] -= 1
        self._launch_ready(force=False)

    def _launch_ready(self, force: bool) -> None:
        """Collectives are issued in ONE order on every rank -- last chunk first, the order the backward pass
        completes them -- whatever order the gradients of this rank arrive in (a rank that discarded its
        batch has no backward pass at all): a chunk goes out once every chunk behind it has gone out and
        its own gradients are complete (or ``force``: the backward pass is over)."""
        while self._next >= 0 and (force or self._pending[self._next] == 0):
            i = self._next
            if not self._packed[i]:
                if not force and self.flat.is_cuda:
                    # weight gradients whose products wait for the end of the pass are adopted, not filled yet:
                    # issue what is queued before the chunk is read
                    from . import ops
                    ops.flush_weight_grads(final=False)
                self._pack_chunk(i)
            self._reduce(i, async_op=not force)
            self._next -= 1

    def _pack_chunk(self, i: int) -> None:
        a, b, params = self.chunks[i]
        piece = self.flat[a:b]
        views = [self._views[self._index[id(p)]] for p in params]
        # a gradient that is still the view handed back by the last unpack (zeroed in place, never replaced:
        # a rank that discarded its batch) already sits where it belongs
        todo = [(v, p.grad) for v, p in zip(views, params)
                if not (p.grad is not None and p.grad.data_ptr() == v.data_ptr() and p.grad.shape == v.shape)]
        missing = [v for v, g in todo if g is None]
        todo = [(v, g) for v, g in todo if g is not None]
        if todo:  # one multi-tensor copy
            torch._foreach_copy_([v for v, _ in todo], [g for _, g in todo])
        if missing:
            torch._foreach_zero_(missing)
        if self._bad is not None:
            piece.masked_fill_(self._bad, 0.0)
        if i == len(self.chunks) - 1:  # the contribution flag travels with the last chunk
            if self._bad is not None:
                self.flat[self.total:] = 1.0 - self._bad.to(self.flat.dtype)
            else:
                self.flat[self.total:] = 1.0
        self._packed[i] = True

    def _span(self, i: int) -> torch.Tensor:
        a, b, _ = self.chunks[i]
        return self.flat[a:b + 1] if i == len(self.chunks) - 1 else self.flat[a:b]

    def _reduce(self, i: int, async_op: bool) -> None:
        self._reduce_piece(self._span(i), async_op)

    def _reduce_piece(self, piece: torch.Tensor, async_op: bool) -> None:
        if _is_rccl(self.group):  # RCCL averages inside the collective
            work = dist.all_reduce(piece, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        elif piece.is_cuda:  # CPU backend with device tensors: staged through host memory
            _all_reduce_sum_via_host(piece, self.group)
            piece.div_(dist.get_world_size(self.group))
            work = None
        else:
            work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if not async_op:
                piece.div_(dist.get_world_size(self.group))
            else:
                work = (work, piece)
        if async_op and work is not None:
            self._work.append(work)

    def pack(self, bad: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Packs every chunk that is not packed yet (nothing is reduced); returns the gradient part of the
        buffer."""
        if not self._armed:
            self.begin(bad)
        for i in range(len(self.chunks)):
            if not self._packed[i]:
                self._pack_chunk(i)
        return self.flat[:self.total]

    def gradients(self) -> torch.Tensor:
        """The buffer's gradients without the alignment gaps, concatenated in parameter order (a copy)."""
        self._buffer()
        return torch.cat([v.reshape(-1) for v in self._views])

    def unpack(self) -> None:
        for p, v in zip(self.params, self._views):
            p.grad = v

    def finish(self, bad: Optional[torch.Tensor] = None) -> torch.Tensor:
        """After the backward pass: reduces whatever has not been handed to a collective yet, waits for all
        of them and installs the averaged gradients.  Returns the share of ranks that contributed (a device
        scalar; 0 means that every rank discarded its batch)."""
        if not self._armed:
            self.begin(bad)
        if not self.overlap:
            # nothing goes out during the backward pass: the whole buffer, contribution flag included, in ONE
            # collective -- the payload is latency-bound, so chunks would only add latencies.  (With overlap a
            # rank without a backward pass must still issue the same chunked collectives as the others.)
            self.pack(bad)
            self._reduce_piece(self.flat, async_op=False)
            self._next = -1
        else:
            self._launch_ready(force=True)
        for work in self._work:
            if isinstance(work, tuple):  # CPU backend: the sum still has to become a mean
                work[0].wait()
                work[1].div_(dist.get_world_size(self.group))
            else:
                work.wait()
        self._work = []
        self._armed = False
        self.unpack()
        return self.flat[self.total]

    def all_reduce_mean(self, group=None, bad: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.finish(bad)

    def reduce_packed(self) -> None:
        """The collective alone, on a buffer that ``pack`` has filled: ONE blocking all-reduce of the whole buffer
        (all chunks and the contribution flag).  A recorded step calls this between two of its graph segments
        (the packing before it and the update after it are recorded, the collective is not); nothing overlaps
        there, so cutting the buffer into chunks would only add collective latencies."""
        self._reduce_piece(self.flat, async_op=False)


class _DataParallelSync:
    """What ``HiVT.training_step`` calls around each backward pass: ``begin`` before it (arms the bucket's
    hooks), the object itself after it (returns the share of contributing ranks)."""

    def __init__(self, trainer: "GanTrainer") -> None:
        self.trainer = trainer

    def begin(self, group: str, bad: Optional[torch.Tensor] = None) -> None:
        self.trainer.buckets[group].begin(bad)

    def __call__(self, group: str, bad: Optional[torch.Tensor] = None):
        return self.trainer._sync(group, bad)


# scalar entries of the step's logs, in the order they are packed for the one reduction per step
_LOG_KEYS = ("reg_loss", "cls_loss", "loss", "d_loss", "g_adv", "g_total", "d_loss_short", "r1_short", "d_loss_mid",
             "r1_mid", "d_loss_long", "r1_long", "g_loss_short", "g_loss_mid", "g_loss_long", "skipped")


class GanTrainer:
    """Owns the optimizers, LR schedule and gradient buckets of one rank.

    ``edge_capacity`` (three integers: agent-agent over all history steps, lane-actor, last-step
    interaction; optionally a fourth: unpadded (step, agent) nodes of the agent-agent stage, without which
    that stage is evaluated for every node) switches the step to the read-back-free mode described in the
    module docstring.  A batch whose filtered edge lists or live nodes do not fit is discarded on the device
    and counted in ``overflowed_steps()``."""

    def __init__(self, model: HiVT, distributed: Optional[bool] = None,
                 edge_capacity: Optional[Tuple[int, ...]] = None, overlap: Optional[bool] = None,
                 sync_logs: bool = True) -> None:
        self.model = model
        self.distributed = (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) \
            if distributed is None else distributed
        self.edge_capacity = tuple(int(c) for c in edge_capacity) if edge_capacity is not None else None
        self.sync_logs = bool(sync_logs)
        self.defer_weight_grads = True
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
            overlap = _is_rccl() if overlap is None else bool(overlap)
            gen = model.generator_parameters() if model.use_gan else list(model.parameters())
            # chunks in the order the backward pass finishes them: decoder, interaction module, lane
            # and temporal stages, agent-agent stage (the buffer keeps the optimizer's parameter order,
            # local encoder first, so the chunk list is walked from the back of the buffer)
            le = model.local_encoder
            chunks = [list(le.aa_encoder.parameters()), list(le.temporal_encoder.parameters())
                      + list(le.al_encoder.parameters()), list(model.global_interactor.parameters()),
                      list(model.decoder.parameters())]
            if not model.use_gan or [id(p) for c in chunks for p in c] != [id(p) for p in gen if p.requires_grad]:
                chunks = None
            self.buckets["generator"] = FlatGradBucket(gen, chunks=chunks, overlap=overlap)
            if model.use_gan:
                self.buckets["critic"] = FlatGradBucket(model.critic_parameters(), overlap=overlap)
        self._overflows: Optional[torch.Tensor] = None
        self._sync_override = None  # a recorded step installs its own reducer here (hivt_gan_amd/graph.py)

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def calibrate(model: HiVT, batches: Iterable[TemporalData], slack: float = 1.1, multiple: int = 256,
                  node_slack: float = 1.05) -> BatchSpec:
        """A ``BatchSpec`` covering ``batches`` (GPU tensors): tensor capacities from their shapes and
        filtered-edge capacities from building their edge sets once (one read-back per batch).  ``slack`` /
        ``node_slack``: head room over the largest edge / node counts seen, for batches that were not among
        ``batches`` (every padded row is computed like a real one, so head room costs its share of the step; a
        batch that does not fit is discarded and flagged)."""
        from .local_encoder import build_edge_plan
        batches = list(batches)
        spec = BatchSpec.covering(batches, slack=node_slack)
        worst = [0, 0, 0, 0]
        for b in batches:
            plan = build_edge_plan(b, model.historical_steps, model.local_encoder.local_radius)
            for i, es in enumerate((plan.agent_agent, plan.lane_actor, plan.interaction)):
                worst[i] = max(worst[i], es.num_edges)
            # fourth capacity: unpadded (step, agent) nodes, the rows the agent-agent stage is evaluated for
            worst[3] = max(worst[3], plan.aa_rows.capacity if plan.aa_rows is not None
                           else model.historical_steps * int(b.num_nodes))
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

    def _sync(self, group: str, bad: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.buckets[group].finish(bad)

    def reduce_logs(self, logs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """The step's scalar logs averaged over the ranks that produced them, in ONE all-reduce (the
        reference logs every key with ``sync_dist=True``, one reduction each, ``models/hivt.py:332-340``).
        A rank that skipped its batch takes part with weight zero, so every rank enters the collective."""
        ref = next(iter(logs.values()))
        vals = torch.zeros(2 * len(_LOG_KEYS), dtype=torch.float32, device=ref.device)
        present = [i for i, k in enumerate(_LOG_KEYS) if k in logs and logs[k].numel() == 1]
        if present:
            vals[present] = torch.stack([logs[_LOG_KEYS[i]].detach().float().reshape(()) for i in present])
            vals[[len(_LOG_KEYS) + i for i in present]] = 1.0
        if _is_rccl() or not vals.is_cuda:
            dist.all_reduce(vals, op=dist.ReduceOp.SUM)
        else:
            _all_reduce_sum_via_host(vals)
        n = len(_LOG_KEYS)
        out = dict(logs)
        for i, k in enumerate(_LOG_KEYS):
            if k in logs:
                out[k] = vals[i] / vals[n + i].clamp(min=1.0)
        return out

    def _weight_grad_scope(self):
        """Small weight-gradient products of a backward pass are issued together at its end; under the overlapped
        data-parallel reduction, whose grad-ready hooks read gradients while the pass is still running, together
        whenever a chunk of parameters is complete (``FlatGradBucket._launch_ready``)."""
        from . import ops
        if self.defer_weight_grads and next(self.model.parameters()).is_cuda:
            return ops.deferred_weight_grads()
        return contextlib.nullcontext()

    def step(self, data) -> Dict[str, torch.Tensor]:
        sync = (self._sync_override or _DataParallelSync(self)) if self.distributed else None
        if self.edge_capacity is None:
            with self._weight_grad_scope():
                logs = self.model.training_step(data, self.opt_g, self.opt_d, sync)
            return self.reduce_logs(logs) if (self.distributed and self.sync_logs) else logs
        data["edge_capacity"] = self.edge_capacity
        with self._weight_grad_scope():
            logs = self.model.training_step(data, self.opt_g, self.opt_d, sync)
        if self.distributed and self.sync_logs:
            logs = self.reduce_logs(logs)
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
