"""Throughput of the G + D + R1 training step (scenes/sec) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One rank per GPU (RCCL all-reduce of the flat gradient buckets); every rank draws its own synthetic
nuScenes-shaped batches of ``--scenes`` scenes (weak scaling).  A step is the whole of
``HiVT.training_step``: device-side edge construction, generator forward, fused supervised loss,
critic update with hinge + R1, generator update.  Rank 0 prints ONE JSON line.

Precision: the named workload is the bf16 one, so by default the dense products take bf16 operands
with fp32 accumulation (everything else fp32; in memory only gradients that nothing but those products
read are bfloat16); ``--matmul fp32`` runs the step in fp32 throughout.  The other setting is timed after
the headline measurement and reported as ``matmul_variant``.

The step is recorded as a HIP graph on fixed shapes and replayed.  A rank cycles through a pool of ``--pool``
distinct batches; the pool is spread over at most ``--max-recordings`` recorded shapes
(``BatchSpec.shape_buckets``), each with ``--capacity-slack`` head room over its largest batch -- the head room
``GanTrainer.calibrate`` gives by default, for batches it has not seen -- and every batch replays the recording
of its bucket (padded rows are computed like real ones).  A batch that did not fit would be discarded and counted
in ``config.discarded_steps``.  ``config.tuned_to_pool_variant`` is the same step on four batches at capacities
fitted to exactly those batches (no head room).  Data parallel: batch i of every rank is padded to the largest
capacities batch i has on any rank, so all ranks cut the same buckets and make the same recordings.
``--no-buckets``: one recording at the pool's largest shape.

The ``roofline`` object describes the edge-attention launch kind that takes the most time per step:
algorithmic bytes of one launch divided by its average duration (the kernel's own start / stop time
stamps).  Every launch kind is listed under ``edge_attention_kernels`` with, beside its fraction of
8 TB/s, the time a plain device copy of the same byte count takes on this GPU in this run
(``copy_same_bytes_us``; ``rate_vs_copy`` > 1 means the attention launch moves its bytes faster than
that copy).  ``dense_roofline`` prices the longest dense product of the step alone (the lane-actor edge rows
through a 128 x 128 layer, forward and weight gradient): flop rate against the dense matrix-core peak of the
operand type and byte rate against HBM.  ``cpu_baseline`` times the CPU oracle's restatement of the same step on a bounded sample
of the same workload (rank 0, single-GPU runs only).
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import hivt_gan_amd as hv  # noqa: E402
from hivt_gan_amd import ops  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak


def model_config(embed_dim: int, local_radius: float = 50.0):
    return dict(historical_steps=20, future_steps=30, num_modes=6, rotate=True, node_dim=2, edge_dim=2,
                embed_dim=embed_dim, num_heads=8, dropout=0.1, num_temporal_layers=4, num_global_layers=3,
                local_radius=local_radius, parallel=False, lr=5e-4, weight_decay=1e-4, T_max=64, use_gan=True,
                lambda_adv=0.1, lambda_r1=1.0, critic_steps=1, short_horizon=10, mid_horizon=20, critic_lr=1e-4)


def seg_attn_bytes(kind: str, n: int, e: int, d: int, h: int, node_terms: bool, n_live: int = None,
                   grad_bytes: int = 4) -> float:
    """Algorithmic HBM bytes of one launch (fp32 rows, int32 CSR).  ``n_live`` = target rows with at
    least one edge: a row without edges needs neither its query nor (backward) its output / output
    gradient, and the kernels skip those loads, so they are not credited.  ``grad_bytes``: bytes per element
    of the dK / dV rows a backward launch writes (2 where they are kept as bfloat16)."""
    s = 4
    n_live = n if n_live is None else n_live
    if kind == "fwd":
        # K, V rows (+ source ids); rowptr; q of live rows; out + saved max / denominator of every row
        b = e * 2 * d * s + n * 4 + n_live * d * s + n * d * s + n * h * 8
        if node_terms:
            b += n * 2 * d * s + e * 4
    else:
        # K, V read, dK, dV written; rowptr; q, out, d_out (+ max / denominator) of live rows; dq of every row
        b = e * 2 * d * (s + grad_bytes) + n * 4 + n_live * (3 * d * s + h * 8) + n * d * s
        if node_terms:
            b += n * 2 * d * s + e * 4
    return float(b)


def copy_same_bytes_us(nbytes: float, device, repeats: int = 15) -> float:
    """Median microseconds of a plain device copy that moves ``nbytes`` in total (half read, half written):
    the rate this device delivers to the simplest streaming kernel at the byte count of one launch, which
    for launches of 50-300 MB is well under the 8 TB/s the roofline is quoted against."""
    n = max(int(nbytes) // 8, 1)
    src = torch.empty(n, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    times = []
    for i in range(repeats + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        e1.synchronize()
        if i >= 3:
            times.append(e0.elapsed_time(e1) * 1e3)
    return sorted(times)[len(times) // 2]


def measured_traffic(embed_dim: int, scenes: int):
    """HBM bytes per launch from the committed counter passes (profiles/*_seg_attn_traffic.json, latest
    file whose workload matches), keyed "kind/caller"; empty when none applies."""
    import glob
    best = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_seg_attn_traffic.json"))):
        with open(path) as f:
            doc = json.load(f)
        if doc.get("workload") == {"embed_dim": embed_dim, "scenes_per_gpu": scenes}:
            best = {k: v["hbm_bytes"] for k, v in doc["per_launch"].items()}
    return best


class _Phases:
    """Exclusive wall time per phase of the CPU step: the oracle's functions are wrapped in place, nested
    calls are charged to the innermost phase."""

    def __init__(self):
        import collections
        self.seconds = collections.defaultdict(float)
        self._stack = []
        self._undo = []

    def wrap(self, owner, name, phase):
        orig = getattr(owner, name)

        def timed_call(*a, **k):
            self._stack.append([time.perf_counter(), 0.0])
            try:
                return orig(*a, **k)
            finally:
                start, child = self._stack.pop()
                elapsed = time.perf_counter() - start
                self.seconds[phase] += elapsed - child
                if self._stack:
                    self._stack[-1][1] += elapsed

        setattr(owner, name, timed_call)
        self._undo.append((owner, name, orig))

    def restore(self):
        for owner, name, orig in reversed(self._undo):
            setattr(owner, name, orig)
        self._undo = []


MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.0}  # dense matrix-core peaks (no sparsity)


def dense_roofline(dev, rows: int, width: int, matmul: str, launches: int = 20):
    """The step's longest dense product priced alone on ``rows`` x ``width`` fp32 activations: the forward launch
    ``y = x W^T + b`` and the weight-gradient call ``dW = dy^T x, db = column sums`` (one grouped call of one term),
    each timed over ``launches`` back-to-back launches between two events on the stream they run on.  Flops are
    the product's own 2 * rows * width^2 (the bf16 forward runs three matrix instructions per product on split
    operands; that is not counted); bytes are the fp32 operands read and written once."""
    x = torch.randn(rows, width, device=dev)
    w = torch.randn(width, width, device=dev) * width ** -0.5
    b = torch.zeros(width, device=dev)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    flops = 2.0 * rows * width * width
    nbytes = {"forward": 4.0 * (2 * rows * width + width * width + width),
              "weight_gradient": 4.0 * (2 * rows * width + width * width + width)}
    out = {"rows": rows, "k": width, "n": width, "operands": matmul, "peak_tflops": MFMA_PEAK_TFLOPS[matmul],
           "peak_gbs": HBM_PEAK_GBS, "launches": launches}
    with torch.no_grad():
        y = ops.fused_linear(x, w, b)
        calls = {"forward": lambda: ops.fused_linear(x, w, b),
                 "weight_gradient": lambda: ops.linear_wgrad_group([(y, x, dw, db, False, False)])}
        for name, call in calls.items():
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(launches):
                call()
            e1.record()
            e1.synchronize()
            sec = e0.elapsed_time(e1) * 1e-3 / launches
            out[name] = {"us": sec * 1e6, "tflops": flops / sec / 1e12, "gbs": nbytes[name] / sec / 1e9,
                         "frac_of_mfma_peak": flops / sec / 1e12 / MFMA_PEAK_TFLOPS[matmul],
                         "frac_of_hbm_peak": nbytes[name] / sec / 1e9 / HBM_PEAK_GBS}
    out["bound"] = "hbm"  # 64 flop per byte of fp32 operand at width 128: below either ridge point
    return out


def cpu_baseline(embed_dim: int, scenes: int, steps: int, warmup: int = 3):
    """The CPU oracle's G + D + R1 step (forward, both backwards, both AdamW updates) on a bounded
    sample of the same workload; returns scenes/sec on the host cores torch is allowed to use and the
    time per phase (edge construction, the three edge-attention stages, temporal encoder, decoder, losses,
    backward passes, optimizers)."""
    from oracle import hivt_oracle as O
    cfg = model_config(embed_dim)
    torch.manual_seed(2022)
    model = hv.HiVT(**cfg)
    full = model.state_dict()
    gen_sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in full.items()
              if not k.startswith("D_")}
    cr_sd = {}
    for k, v in full.items():
        if k.startswith("D_"):
            t = v.detach().clone()
            cr_sd[k] = t.requires_grad_(True) if (k.endswith("weight_orig") or k.endswith("bias")) else t
    gen_params = [v for v in gen_sd.values() if v.requires_grad]
    cr_params = [v for v in cr_sd.values() if v.requires_grad]
    opt_g = torch.optim.AdamW(gen_params, lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    opt_d = torch.optim.AdamW(cr_params, lr=cfg["critic_lr"], weight_decay=0.0)
    ocfg = dict(cfg)
    phases = _Phases()
    for name in ("temporal_edges", "distance_drop_edge", "subgraph"):
        phases.wrap(O, name, "edge_build")
    for name, phase in (("aa_encoder", "aa_encoder"), ("temporal_encoder", "temporal_encoder"), ("al_encoder", "al_encoder"),
                        ("global_interactor", "global_interactor"), ("mlp_decoder", "decoder"),
                        ("supervised_losses", "losses"), ("discriminator_loss", "losses"),
                        ("generator_adv_loss", "losses")):
        phases.wrap(O, name, phase)
    phases.wrap(torch.autograd, "grad", "backward")
    phases.wrap(opt_g, "step", "optimizers")
    phases.wrap(opt_d, "step", "optimizers")

    def critic_update(grads):
        for p, g in zip(cr_params, grads.values()):
            p.grad = g
        opt_d.step()

    ocfg["critic_update"] = critic_update
    times = []
    try:
        for i in range(steps + warmup):
            batch = hv.make_batch(scenes, first_index=50_000 + i * scenes)
            data = {k: batch[k] for k in batch.keys()}
            if i == warmup:
                phases.seconds.clear()  # phases of the timed steps only
            t0 = time.perf_counter()
            res = O.gan_step_grads(gen_sd, cr_sd, data, ocfg, training=True)
            for p, k in zip(gen_params, [k for k, v in gen_sd.items() if v.requires_grad]):
                p.grad = res["g_grads"][k]
            opt_g.step()
            times.append(time.perf_counter() - t0)
    finally:
        phases.restore()
    med = statistics.median(times[warmup:])
    per_phase = {k: round(v / steps * 1e3, 1) for k, v in sorted(phases.seconds.items(), key=lambda kv: -kv[1])}
    per_phase["other"] = round(max(sum(times[warmup:]) / steps * 1e3 - sum(per_phase.values()), 0.0), 1)
    return {"value": scenes / med, "unit": "scenes/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed steps (+{warmup} warm-up) of the CPU oracle's G+D+R1 step incl. AdamW on {scenes} "
                      f"synthetic scenes per step, HiVT-{embed_dim} fp32, median {med * 1e3:.0f} ms/step",
            "phase_ms_per_step": per_phase}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--embed-dim", type=int, default=128)
    ap.add_argument("--scenes", type=int, default=128, help="scenes per GPU per step")
    ap.add_argument("--pool", type=int, default=32, help="distinct synthetic batches cycled per rank")
    ap.add_argument("--max-recordings", type=int, default=4,
                    help="recorded shapes the pool's batches are spread over (each batch replays the smallest that holds it)")
    ap.add_argument("--no-pool-fitted", action="store_true",
                    help="skip the extra timing of four batches at capacities fitted to exactly those batches "
                         "(reported as config.tuned_to_pool_variant, never as value)")
    ap.add_argument("--dense", action="store_true",
                    help="dense-scene stress shape instead of the headline workload: 200 agents and 1000 lane "
                         "segments per scene, 150 m radius, every step valid (use with --scenes 2..8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-buckets", dest="buckets", action="store_false",
                    help="one recording at the pool's largest shape for every batch (default: one recording per distinct "
                         "padded shape of the pool)")
    ap.add_argument("--capacity-slack", type=float, default=1.1,
                    help="head room of the recorded step's fixed edge capacities over the largest batch of a shape bucket "
                         "(nodes: at most 1.05); the default is GanTrainer.calibrate's own, what a run over a data stream "
                         "uses -- padded rows are computed like real ones, so head room costs its share of the step")
    ap.add_argument("--matmul", choices=("fp32", "bf16"), default="bf16",
                    help="operand precision of the dense products: bf16 operands with fp32 accumulation (the named "
                         "workload's precision, results within 2e-2) or fp32 throughout (within 1e-5)")
    ap.add_argument("--no-variant", action="store_true",
                    help="skip the extra timing of the step with the other --matmul precision")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the eager steps that price the edge-attention kernels after the timed region (roofline "
                         "null; for kernel traces whose last steps should be the timed replays)")
    ap.add_argument("--no-tune", action="store_true", help="keep the default launch schedule (no timing of alternatives)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every step eagerly instead of replaying the recorded step (single-GPU runs record the "
                         "step into a HIP graph by default; data-parallel runs always launch eagerly)")
    ap.add_argument("--host-guards", action="store_true",
                    help="evaluate the NaN / finite guards and edge counts on the host (three read-backs per step)")
    ap.add_argument("--cpu-scenes", type=int, default=128, help="scenes per step of the CPU baseline (the GPU step's batch)")
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed steps of the CPU baseline (median reported)")
    ap.add_argument("--cpu-warmup", type=int, default=3, help="untimed steps of the CPU baseline before them")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RCCL is the data path; HIVT_DIST_BACKEND=gloo exists to rehearse the multi-rank control flow on a
    # box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("HIVT_DIST_BACKEND", "nccl")
    local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if args.gpus != world:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    ops.set_matmul_precision(args.matmul)
    torch.manual_seed(2022)
    scene_kwargs = dict(num_agents=200, num_lanes=1000, all_valid=True, local_radius=150.0) if args.dense else {}
    model = hv.HiVT(**model_config(args.embed_dim, 150.0 if args.dense else 50.0)).to(dev)
    model.train()

    # synthetic batches, resident in HBM before the timed region; distinct scenes per rank
    pool = []
    for i in range(args.pool):
        b = hv.make_batch(args.scenes, first_index=(rank * args.pool + i) * args.scenes, **scene_kwargs).to(dev)
        pool.append((b, b["y"]))
    # capacities of the filtered edge lists, measured once over the pool (--capacity-slack head room): with them
    # the step needs no host read-back (live counts, guards and the overflow flag stay on the device)
    spec = None if args.host_guards else hv.GanTrainer.calibrate(model, [b for b, _ in pool], slack=args.capacity_slack,
                                                                 node_slack=min(args.capacity_slack, 1.05))
    capacity = spec.edge_capacity if spec is not None else None
    trainer = hv.GanTrainer(model, edge_capacity=capacity)
    torch.manual_seed(1000 + rank)  # dropout streams differ per rank

    def eager_step(i: int):
        batch, y_raw = pool[i % len(pool)]
        batch["y"] = y_raw            # forward rotates the targets in place on the batch object
        batch["edge_plan"] = None     # edge construction belongs to the step
        return trainer.step(batch)

    # launch schedule chosen by measurement (untimed set-up, like the capacity calibration above)
    tuned = trainer.tune(eager_step) if not args.no_tune else {}
    graphed, graph_note = None, None
    if spec is not None and not args.no_graph:
        # the whole step recorded once and replayed: batches padded to one fixed shape (inert padding), each
        # step copies its batch into the recording's input tensors (device to device) and launches the graph
        # (data parallel: a few graph segments with the gradient all-reduces between them)
        from hivt_gan_amd.graph import GraphedStep
        if world > 1:  # every rank pads to the same shapes: the largest capacities of any rank
            caps = torch.tensor([spec.num_nodes, spec.num_lanes, spec.num_edges, spec.num_lane_actor, *spec.edge_capacity],
                                dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(caps, op=dist.ReduceOp.MAX)
            c = [int(v) for v in caps.tolist()]
            spec = hv.BatchSpec(c[0], c[1], c[2], c[3], edge_capacity=tuple(c[4:]))
            trainer.edge_capacity = spec.edge_capacity
        from hivt_gan_amd.graph import pack_inputs
        padded = [pack_inputs(hv.pad_batch(b, spec)) for b, _ in pool]  # one buffer per batch: one copy per step
        try:
            graphed = GraphedStep(trainer, spec, padded[0], critic_streams=bool(tuned.get("critic_streams", False)))
        except Exception as exc:  # noqa: BLE001 -- the eager step is always available
            graph_note = f"recording failed ({type(exc).__name__}: {exc}); launching eagerly"
            if rank == 0:
                print("warning: " + graph_note, file=sys.stderr)
        if world > 1:  # all ranks replay or none does
            ok = torch.tensor([1 if graphed is not None else 0], device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                graphed = None

    # Shape buckets: every padded row of a fixed-shape recording is computed like a real one, so a batch replays the
    # recording of the smallest shape that holds it.  The pool's batches are spread over at most --max-recordings
    # shapes (BatchSpec.shape_buckets: the cuts that leave the fewest padded rows), each with --capacity-slack head
    # room over its largest batch.  All recordings share the model, the optimizers and their state.  If any of them
    # cannot be made, the one recording at the pool's largest shape above serves every batch.
    def record_buckets(batches, slack: float, max_recordings: int, first_seed: int):
        """(recordings, bucket of every batch, every batch padded to its bucket's shape)."""
        from hivt_gan_amd.graph import GraphedStep, pack_inputs
        own = [hv.GanTrainer.calibrate(model, [b], slack=slack, node_slack=min(slack, 1.05)) for b in batches]
        if world > 1:
            # batch i of every rank at the same capacities (the largest of any rank): every rank then cuts the same
            # buckets and makes the same number of recordings, whose warm-up steps hold collectives
            caps = torch.tensor([sp.key() for sp in own], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(caps, op=dist.ReduceOp.MAX)
            own = [hv.BatchSpec(c[0], c[1], c[2], c[3], edge_capacity=tuple(c[4:])) for c in caps.tolist()]
        shapes, which = hv.BatchSpec.shape_buckets(own, max_recordings)
        examples = [batches[which.index(j)] for j in range(len(shapes))]
        graphs = [GraphedStep(trainer, sp, pack_inputs(hv.pad_batch(ex, sp)), seed_offset=first_seed + j,
                              critic_streams=bool(tuned.get("critic_streams", False)))
                  for j, (sp, ex) in enumerate(zip(shapes, examples))]
        return graphs, which, [pack_inputs(hv.pad_batch(b, shapes[j])) for b, j in zip(batches, which)]

    def all_ranks(ok: bool) -> bool:
        if world == 1:
            return ok
        flag = torch.tensor([1 if ok else 0], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    bucket_of, bucket_graphs, bucket_padded = None, [], []
    if graphed is not None and args.buckets:
        try:
            bucket_graphs, bucket_of, bucket_padded = record_buckets([b for b, _ in pool], args.capacity_slack,
                                                                     args.max_recordings, 1)
        except Exception as exc:  # noqa: BLE001 -- the single recording stays in use
            if rank == 0:
                print(f"warning: shape buckets not recorded ({type(exc).__name__}: {exc})", file=sys.stderr)
            bucket_of, bucket_graphs, bucket_padded = None, [], []
        if not all_ranks(bucket_of is not None):  # all ranks replay their own recordings or none does
            bucket_of, bucket_graphs, bucket_padded = None, [], []

    def run_step(i: int):
        if bucket_of is not None:
            j = i % len(bucket_padded)
            return bucket_graphs[bucket_of[j]].run(bucket_padded[j])
        if graphed is not None:
            return graphed.run(padded[i % len(padded)])
        return eager_step(i)

    for i in range(args.warmup):
        run_step(i)
    def timed(first: int, steps: int) -> float:
        """Seconds for ``steps`` steps between two barrier + synchronize brackets, max over ranks."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            timed.logs = run_step(first + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    records = []
    if graphed is None:
        ops.SEG_ATTN_TIMING = records
    elapsed = timed(args.warmup, args.steps)
    ops.SEG_ATTN_TIMING = None
    logs = {k: v.clone() for k, v in timed.logs.items()}
    if graphed is not None and not args.no_kernel_timing:
        # a replayed graph cannot carry the per-kernel time stamps: the edge-attention kernels are priced on
        # the same number of eagerly launched steps of the same trainer right after the timed region
        ops.SEG_ATTN_TIMING = records
        for i in range(args.steps):
            eager_step(args.warmup + i)
        torch.cuda.synchronize()
        ops.SEG_ATTN_TIMING = None

    # What the head room and the few shapes cost: the first four batches of the pool at capacities fitted to exactly
    # those batches (no head room, one recording per distinct shape), timed the same way (reported beside the
    # headline, never as ``value``)
    fitted = None
    if bucket_of is not None and not args.no_pool_fitted:
        headline = (bucket_graphs, bucket_of, bucket_padded)
        try:
            few = [b for b, _ in pool[:4]]
            bucket_graphs, bucket_of, bucket_padded = record_buckets(few, 1.0, len(few), 101)
            for i in range(3):
                run_step(i)
            fitted = {"ms_per_step": timed(0, args.steps) / args.steps * 1e3, "pool": len(few), "capacity_slack": 1.0,
                      "recordings": len(bucket_graphs), "nodes": [g.spec.num_nodes for g in bucket_graphs]}
            fitted["value"] = world * args.scenes / fitted["ms_per_step"] * 1e3
        except Exception as exc:  # noqa: BLE001 -- an extra, never the headline
            if rank == 0:
                print(f"warning: pool-fitted variant not timed ({type(exc).__name__}: {exc})", file=sys.stderr)
            fitted = None
        bucket_graphs, bucket_of, bucket_padded = headline
        if not all_ranks(fitted is not None):
            fitted = None

    # the same step with the other operand precision of the dense products, timed the same way after the
    # headline measurement (reported beside it, never as ``value``)
    other = None
    if not args.no_variant:
        name = "bf16" if args.matmul == "fp32" else "fp32"
        ops.set_matmul_precision(name)
        if graphed is not None:
            graphed.record()  # the operand precision is part of what a recording fixes
        for g in bucket_graphs:
            g.record()
        for i in range(3):
            run_step(i)
        other = (name, timed(args.warmup + args.steps, args.steps))
        ops.set_matmul_precision(args.matmul)

    if rank == 0:
        groups = {}
        for kind, tag, n, rowptr, d, h, node_terms, ev0, ev1, grad_bytes in records:
            rp = rowptr.cpu()
            groups.setdefault((kind, tag, node_terms, d, h, grad_bytes), []).append(
                (n, int(rp[n]), ev0.elapsed_time(ev1) * 1e-3, int((rp[1:] > rp[:-1]).sum())))
        roofline, extra = None, []
        traffic = measured_traffic(args.embed_dim, args.scenes)
        for (kind, tag, node_terms, d, h, grad_bytes), rows in groups.items():
            by = sum(seg_attn_bytes(kind, n, e, d, h, node_terms, live, grad_bytes) for n, e, _, live in rows) / len(rows)
            tm = sum(r[2] for r in rows) / len(rows)
            entry = {"kernel": f"seg_attn_{kind}_kernel", "caller": tag, "bound": "hbm", "achieved": by / tm / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / tm / 1e9 / HBM_PEAK_GBS, "traffic": traffic.get(f"{kind}/{tag}"),
                     "bytes_per_launch": by, "avg_launch_us": tm * 1e6, "launches": len(rows),
                     "grad_row_dtype": "bf16" if grad_bytes == 2 else "f32",
                     "edges_per_launch": sum(r[1] for r in rows) / len(rows),
                     "rows_per_launch": sum(r[0] for r in rows) / len(rows),
                     "rows_with_edges_per_launch": sum(r[3] for r in rows) / len(rows)}
            entry["us_per_step"] = tm * 1e6 * len(rows) / args.steps
            entry["copy_same_bytes_us"] = copy_same_bytes_us(by, dev)
            entry["rate_vs_copy"] = entry["copy_same_bytes_us"] / entry["avg_launch_us"]
            extra.append(entry)
        # dominant kernel = the edge-attention launch kind that takes the most time per step
        if extra:
            roofline = dict(max(extra, key=lambda en: en["us_per_step"]))
            tot_b = sum(en["bytes_per_launch"] * en["launches"] for en in extra)
            tot_t = sum(en["avg_launch_us"] * 1e-6 * en["launches"] for en in extra)
            roofline["all_kinds_time_weighted_frac"] = tot_b / tot_t / 1e9 / HBM_PEAK_GBS
        result = {
            "metric": "train scenes/sec (G+D+R1 step)", "value": world * args.scenes * args.steps / elapsed,
            "unit": "scenes/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.matmul == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": f"HiVT-{args.embed_dim} full G+D+R1 train step, {args.scenes} synthetic "
                                   + ("dense (200 agents, 1000 lane segments, 150 m radius) " if args.dense else
                                      "nuScenes-shaped ")
                                   + "scenes per GPU per step, dropout 0.1, AdamW x2, "
                                   + ("bf16 dense products" if args.matmul == "bf16" else "fp32"),
                       "scenes_per_gpu": args.scenes, "embed_dim": args.embed_dim, "parallelism": f"dp{world}",
                       "guards": "host" if args.host_guards else "device",
                       "launch": ("recorded HIP graph, %d segment(s) per step (%d kernel + %d copy nodes), inputs copied "
                                  "into its static tensors" % (sum(1 for k, _ in graphed.segments if k == "graph"),
                                                               graphed.nodes.get("kernel", 0),
                                                               graphed.nodes.get("memcpy", 0))
                                  + ("; the pool's %d batches spread over %d recorded shapes (nodes %s, batches per shape %s)"
                                     % (len(pool), len(bucket_graphs), [g.spec.num_nodes for g in bucket_graphs],
                                        [bucket_of.count(j) for j in range(len(bucket_graphs))])
                                     if bucket_of is not None else ""))
                       if graphed is not None else (graph_note or "eager"),
                       "kernel_timing": "kernel time stamps of %d eager steps after the timed region" % args.steps
                       if graphed is not None else "kernel time stamps inside the timed region",
                       "discarded_steps": trainer.overflowed_steps(), "tuned": tuned,
                       "pool": len(pool), "backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
                       "rccl_ranks": dist.get_world_size() if world > 1 and backend == "nccl" else 0,
                       "capacities": None if spec is None else
                       {"slack_over_largest_batch": args.capacity_slack, "node_slack": min(args.capacity_slack, 1.05),
                        "largest_batch_nodes": max(int(b.num_nodes) for b, _ in pool),
                        "smallest_batch_nodes": min(int(b.num_nodes) for b, _ in pool),
                        "recorded_shapes": [{"nodes": g.spec.num_nodes, "batches": bucket_of.count(j),
                                             "filtered_edges_aa_al_interaction_aa_rows": list(g.spec.edge_capacity)}
                                            for j, g in enumerate(bucket_graphs)] if bucket_of is not None else
                        [{"nodes": spec.num_nodes, "batches": len(pool),
                          "filtered_edges_aa_al_interaction_aa_rows": list(spec.edge_capacity)}]},
                       "tuned_to_pool_variant": fitted,
                       "precision": "fp32 throughout" if args.matmul == "fp32" else
                       "dense products on the bf16 matrix instructions with fp32 accumulation: forward products on "
                       "split (hi + lo) bf16 operands, gradient products on bf16 operands; attention, norms, "
                       "losses, optimizer state and activations fp32 (gradients that only the gradient products read "
                       "are kept as bfloat16, which those products round them to anyway); every parameter gradient within "
                       "2e-2 of the float64 oracle (tests/test_gpu_config3.py)"},
            "roofline": roofline, "edge_attention_kernels": extra,
            "losses": {k: float(v) for k, v in logs.items() if v.numel() == 1},
        }
        if other is not None:
            result["matmul_variant"] = {"matmul": other[0], "ms_per_step": other[1] / args.steps * 1e3,
                                        "value": world * args.scenes * args.steps / other[1], "unit": "scenes/sec",
                                        "note": "same step and schedule, dense products on "
                                                + ("the bf16 matrix instructions (results within 2e-2)"
                                                   if other[0] == "bf16" else "fp32 operands")}
        try:
            al_rows = max((en["edges_per_launch"] for en in extra if en["caller"] == "ALEncoder"), default=0.0)
            result["dense_roofline"] = dense_roofline(dev, int(al_rows) or 136_000, args.embed_dim, args.matmul)
        except Exception as exc:  # noqa: BLE001 -- an extra object of the line
            result["dense_roofline"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.embed_dim, args.cpu_scenes, args.cpu_steps, args.cpu_warmup)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
