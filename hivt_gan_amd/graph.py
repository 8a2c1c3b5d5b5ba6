"""The G + D + R1 step recorded once into a HIP graph and replayed.

An eagerly launched step is about a thousand kernel launches of a few microseconds each, so the
launch thread, not the GPU, bounds it on all but the fastest hosts.  Everything a recorded step needs
is in place in the read-back-free mode of ``GanTrainer`` (``edge_capacity``): tensor shapes are fixed
(``pad_batch``), live edge counts, guards and the overflow flag stay in device memory, the library
launches plain kernels only (no runtime memset / memcpy, no allocation), the optimizers read their step
counter and the skip flag from device memory, and every dropout site adds a device-side seed to its
per-call constant, which the recorded step advances by itself.

A data-parallel step is recorded in segments: the gradient all-reduces run eagerly between them (the
payloads are 1.5 MB and 10 MB, a fraction of a millisecond on xGMI, so what matters for scaling is that
the ranks stay GPU-bound, not that the collectives hide behind the backward pass).

What a recorded step fixes until it is recorded again: the critic-stream schedule and the batch shapes
(``BatchSpec``).  The learning rate is not among them: the library's AdamW kernel reads it from a device scalar that
``run`` refreshes before every replay (``FusedAdamW.sync_lr``); only a recording whose update went through the
framework's kernel (the rate is a launch argument there) is recorded again when the rate changes.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import losses, ops
from .data import BatchSpec, TemporalData, pad_batch

# per-replay advance of the device-side dropout seed: an odd constant unrelated to the multiplier the
# kernels use per element, so that consecutive steps do not see shifted copies of one mask
_SEED_STRIDE = 0x51342543DE82EF95

# tensors of a padded batch that a step reads
_INPUT_KEYS = ("x", "positions", "y", "padding_mask", "bos_mask", "rotate_angles", "lane_vectors", "is_intersections",
               "turn_directions", "traffic_controls", "edge_index", "lane_actor_index", "lane_actor_vectors", "scene_range")


def pack_inputs(batch: TemporalData) -> TemporalData:
    """The tensors of a padded GPU batch that a step reads, re-laid as views of ONE byte buffer (each on a 256-byte
    boundary).  A batch packed this way is handed to a recording with the same layout in a single device copy
    instead of one per tensor.  Returns a new batch; the given one is left alone."""
    keys = [k for k in _INPUT_KEYS if k in batch]
    layout, total = [], 0
    for k in keys:
        t = batch[k]
        layout.append((k, total, tuple(t.shape), t.dtype))
        total += -(-t.numel() * t.element_size() // 256) * 256
    dev = batch[keys[0]].device
    arena = torch.empty(max(total, 256), dtype=torch.uint8, device=dev)
    out = TemporalData(**{k: batch[k] for k in batch.keys() if k not in ("edge_plan", "_prep", "_arena", "_arena_layout")})
    for k, off, shape, dtype in layout:
        n = batch[k].numel() * batch[k].element_size()
        view = arena[off:off + n].view(dtype).view(shape)
        view.copy_(batch[k])
        out[k] = view
    out["_arena"] = arena
    out["_arena_layout"] = tuple(layout)
    return out


_HIP_NODE_KINDS = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event",
                   7: "event_record"}


def node_census(graph: "torch.cuda.CUDAGraph") -> Dict[str, int]:
    """Nodes of a recorded graph (constructed with ``keep_graph=True``) by kind, read through
    hipGraphGetNodes / hipGraphNodeGetType."""
    import collections
    import ctypes
    handle = graph.raw_cuda_graph()
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_size_t(0)
    rc = hip.hipGraphGetNodes(ctypes.c_void_p(handle), None, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"hipGraphGetNodes failed with code {rc}")
    nodes = (ctypes.c_void_p * max(n.value, 1))()
    hip.hipGraphGetNodes(ctypes.c_void_p(handle), nodes, ctypes.byref(n))
    kinds = collections.Counter()
    for node in nodes[:n.value]:
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(t))
        kinds[_HIP_NODE_KINDS.get(t.value, f"type{t.value}")] += 1
    return dict(kinds)


class _RecordedSync:
    """Gradient reduction of a recorded data-parallel step: packing (and zeroing a discarded rank's share)
    is recorded, the collectives run eagerly BETWEEN two graph segments, the hand-back of the averaged
    gradients is host-side bookkeeping."""

    def __init__(self, owner: "GraphedStep") -> None:
        self.owner = owner

    def begin(self, group: str, bad: Optional[torch.Tensor] = None) -> None:
        self.owner.trainer.buckets[group].begin(bad)

    def __call__(self, group: str, bad: Optional[torch.Tensor] = None):
        bucket = self.owner.trainer.buckets[group]
        bucket.pack(bad)
        self.owner._break(bucket.reduce_packed)
        bucket._armed = False
        bucket.unpack()
        return bucket.flat[bucket.total]


class GraphedStep:
    """``trainer.step`` on fixed-shape batches as one graph launch (single process) or as a few graph
    segments with the gradient all-reduces between them (data parallel).

    ``example`` is a batch already padded to ``spec`` and resident on the GPU; copies of its tensors become
    the static inputs of the recording.  ``run(batch)`` copies another padded batch into them (one
    device-to-device copy for a batch prepared with ``pack_inputs``, a dozen otherwise) and replays; ``run()``
    replays on whatever the static inputs hold."""

    def __init__(self, trainer, spec: BatchSpec, example: TemporalData, warmup: int = 3,
                 critic_streams: bool = False, seed_offset: int = 0) -> None:
        if trainer.edge_capacity is None:
            raise RuntimeError("a recorded step needs the read-back-free mode (GanTrainer(edge_capacity=...))")
        self.trainer = trainer
        self.spec = spec
        # the capacities this recording was made for: several recordings of one trainer (shape buckets) each
        # run the step with their own
        self.edge_capacity = tuple(spec.edge_capacity) if spec.edge_capacity is not None else tuple(trainer.edge_capacity)
        # the static inputs live in one byte buffer (pack_inputs), so that a batch laid out the same way arrives
        # in one copy
        self.static = pack_inputs(TemporalData(**{k: (example[k].clone() if torch.is_tensor(example[k]) else example[k])
                                                  for k in example.keys()
                                                  if k not in ("edge_plan", "_prep", "_arena", "_arena_layout")}))
        self._y = self.static["y"]
        dev = self._y.device
        self.device = dev
        # (recordings of one trainer start their dropout streams apart)
        self.seed = torch.full((), int(seed_offset) * 0x5DEECE66D, dtype=torch.int64, device=dev)
        self.critic_streams = bool(critic_streams)
        self.warmup = int(warmup)
        self.nodes: Dict[str, int] = {}
        self.segments: list = []  # ("graph", CUDAGraph) / ("call", function) in replay order
        self.logs: Dict[str, torch.Tensor] = {}
        self._stream = torch.cuda.Stream(device=dev)
        self._capturing = False
        self._pool = None
        self._sync = _RecordedSync(self) if trainer.distributed else None
        if trainer.distributed:
            for bucket in trainer.buckets.values():
                bucket.overlap = False  # the reductions of a recorded step sit between its segments
        self.sync_logs = bool(trainer.distributed and trainer.sync_logs)
        self.replays = 0
        self.record()

    @property
    def graph(self):
        """The recording of a single-process step (one segment)."""
        graphs = [g for kind, g in self.segments if kind == "graph"]
        return graphs[0] if len(graphs) == 1 else None

    # ------------------------------------------------------------------------------------------
    def _one_step(self) -> Dict[str, torch.Tensor]:
        t = self.trainer
        prev = (ops.DEVICE_SEED, ops._seed_calls, losses.CONCURRENT_CRITICS, t._sync_override, t.sync_logs)
        prev_capacity = t.edge_capacity
        ops.DEVICE_SEED, ops._seed_calls = self.seed, 0
        losses.CONCURRENT_CRITICS = self.critic_streams
        t._sync_override, t.sync_logs = self._sync, False
        t.edge_capacity = self.edge_capacity
        try:
            self.seed.add_(_SEED_STRIDE)
            self.static["y"] = self._y       # forward rebinds the entry to the rotated targets
            self.static["edge_plan"] = None  # edge construction belongs to the step
            return t.step(self.static)
        finally:
            ops.DEVICE_SEED, ops._seed_calls, losses.CONCURRENT_CRITICS, t._sync_override, t.sync_logs = prev
            t.edge_capacity = prev_capacity

    def _begin_segment(self) -> None:
        g = torch.cuda.CUDAGraph(keep_graph=True)
        g.capture_begin(pool=self._pool, capture_error_mode="global")
        self.segments.append(("graph", g))

    def _end_segment(self) -> None:
        self.segments[-1][1].capture_end()

    def _break(self, fn) -> None:
        """``fn`` must run outside any recording (a collective): eagerly right away, or, while recording,
        between the segment that ends here and the one that starts here."""
        if not self._capturing:
            fn()
            return
        self._end_segment()
        self.segments.append(("call", fn))
        self._begin_segment()

    def _snapshot(self):
        """Copies of everything a training step changes: parameters, buffers (the critics' power-iteration
        vectors), optimizer state, the dropout seed and the overflow counter."""
        t = self.trainer
        tensors = list(t.model.parameters()) + list(t.model.buffers())
        saved = [(x, x.detach().clone()) for x in tensors]
        seen = {id(x) for x in tensors}
        known = []
        for opt in (t.opt_g, t.opt_d):
            if opt is None:
                continue
            for st in opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v) and id(v) not in seen:
                        seen.add(id(v))
                        saved.append((v, v.detach().clone()))
                        known.append(id(v))
        saved.append((self.seed, self.seed.clone()))
        overflows = t._overflows.clone() if t._overflows is not None else None
        return saved, set(known), overflows

    def _restore(self, snapshot) -> None:
        saved, known, overflows = snapshot
        t = self.trainer
        with torch.no_grad():
            for x, copy in saved:
                x.copy_(copy)
            # optimizer state that the warm-up steps created starts from zero, as a first step would find it
            fresh = {}
            for opt in (t.opt_g, t.opt_d):
                if opt is None:
                    continue
                for st in opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v) and id(v) not in known:
                            fresh[id(v)] = v
            for v in fresh.values():
                v.zero_()
            if t._overflows is not None:
                if overflows is not None:
                    t._overflows.copy_(overflows)
                else:
                    t._overflows.zero_()

    def record(self) -> None:
        """(Re)records the step: ``warmup`` eager steps on the recording stream (allocator, optimizer
        state, autograd nodes settle there), then one step under capture.  The warm-up steps run on whatever the
        static inputs hold, so what they changed -- parameters, optimizer state, power-iteration vectors, the
        dropout seed, the overflow counter -- is put back afterwards: recording leaves the training state as it
        found it (the capture itself executes nothing)."""
        snapshot = self._snapshot()
        current = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(current)
        with torch.cuda.stream(self._stream):
            for _ in range(self.warmup):
                self._one_step()
            self._restore(snapshot)
        current.wait_stream(self._stream)
        torch.cuda.synchronize(self.device)
        self.segments = []
        self._pool = torch.cuda.graph_pool_handle()
        import gc
        gc.collect()
        with torch.cuda.stream(self._stream):
            self._begin_segment()
            self._capturing = True
            try:
                self.logs = self._one_step()
            finally:
                self._capturing = False
                self._end_segment()
        # A runtime memset recorded as a graph node takes no effect when the graph is replayed on this stack
        # (torch 2.10 / ROCm 7.2): a multi-block framework reduction, which clears its semaphores that way,
        # returns wrong sums from the second replay on.  The step is written to issue none; refuse to replay
        # a recording that holds one rather than train on wrong numbers.
        self.nodes = {}
        for kind, g in self.segments:
            if kind == "graph":
                for k, v in node_census(g).items():
                    self.nodes[k] = self.nodes.get(k, 0) + v
        if self.nodes.get("memset", 0):
            self.segments = []
            raise RuntimeError(f"the recorded step holds {self.nodes['memset']} runtime memset node(s) {self.nodes}; "
                               "tools/find_memsets.py names the operators that issue them")
        for kind, g in self.segments:
            if kind == "graph":
                g.instantiate()
        self._lr = [g["lr"] for opt in (self.trainer.opt_g, self.trainer.opt_d) if opt is not None
                    for g in opt.param_groups]
        # every update of the recording went through the library's kernel, which reads the rate from device memory
        opts = [opt for opt in (self.trainer.opt_g, self.trainer.opt_d) if opt is not None]
        self._lr_on_device = all(getattr(opt, "last_step_on_library", False) for opt in opts)

    def load(self, batch: TemporalData) -> None:
        """Copies a batch padded to the same ``BatchSpec`` into the static inputs: one copy for a batch that
        ``pack_inputs`` laid out like them, one copy per tensor otherwise."""
        # the route of the edge construction was fixed when the step was recorded: a recording made on batches whose
        # candidate lists are all ordered pairs builds those pairs itself and never reads `edge_index`, so a batch
        # with another candidate list would silently get the wrong edge set
        if bool(batch.get("all_pairs", False)) != bool(self.static.get("all_pairs", False)):
            raise ValueError("this step was recorded for batches with all_pairs = %s; the batch has all_pairs = %s "
                             "(record a step of its own for such batches, or launch them eagerly)"
                             % (bool(self.static.get("all_pairs", False)), bool(batch.get("all_pairs", False))))
        if batch.get("_arena_layout") is not None and batch["_arena_layout"] == self.static["_arena_layout"]:
            self.static["_arena"].copy_(batch["_arena"], non_blocking=True)
            return
        for k in _INPUT_KEYS:
            if k not in self.static:
                continue
            dst = self._y if k == "y" else self.static[k]
            src = batch[k]
            if src.shape != dst.shape:
                raise ValueError(f"{k}: shape {tuple(src.shape)} does not match the recorded {tuple(dst.shape)}; "
                                 "pad the batch to the recording's BatchSpec")
            dst.copy_(src, non_blocking=True)

    def run(self, batch: Optional[TemporalData] = None) -> Dict[str, torch.Tensor]:
        """One step; returns the recording's loss tensors (overwritten by the next replay)."""
        for opt in (self.trainer.opt_g, self.trainer.opt_d):
            if opt is not None and hasattr(opt, "sync_lr"):
                opt.sync_lr()  # the recorded update reads its learning rate from a device scalar
        lr = [g["lr"] for opt in (self.trainer.opt_g, self.trainer.opt_d) if opt is not None for g in opt.param_groups]
        if lr != self._lr and not self._lr_on_device:  # a launch argument of the recorded update: record again
            self.record()
        if batch is not None:
            self.load(batch)
        for kind, item in self.segments:
            if kind == "graph":
                item.replay()
            else:
                item()
        self.replays += 1
        return self.trainer.reduce_logs(self.logs) if self.sync_logs else self.logs


def padded_pool(batches, spec: BatchSpec):
    """Every batch of ``batches`` (GPU) padded to ``spec`` and packed for one-copy loading."""
    return [pack_inputs(pad_batch(b, spec)) for b in batches]
