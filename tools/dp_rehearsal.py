"""Two data-parallel ranks sharing cuda:0 over gloo: the multi-rank control flow of ``GanTrainer`` with
the real GPU model (RCCL needs one device per rank, so this rehearses everything but the transport).

Each rank logs its stages to gpurun_out/dp_rank<r>.log and dumps its Python stack if it stalls.
Checks: ranks hold identical parameters after every step, the losses are finite, the first step's reduced
generator gradient equals the mean of the two ranks' local gradients, and the step replayed from its recorded
segments (reductions between them) leaves the same bits in the parameters as the eager step from the same state.
"""
import faulthandler
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")


def worker(rank, world, port, steps):
    log = open(os.path.join(OUT, f"dp_rank{rank}.log"), "w", buffering=1)
    faulthandler.dump_traceback_later(60, exit=True, file=log)

    def say(*a):
        print(*a, file=log)

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import hivt_gan_amd as hv
    from bench import model_config
    backend = os.environ.get("HIVT_REHEARSAL_BACKEND", "gloo")
    # RCCL wants a device per rank; over gloo the ranks share cuda:0
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    say("group up", backend)
    torch.manual_seed(2022)
    cfg = model_config(64)
    cfg["dropout"] = 0.0
    model = hv.HiVT(**cfg).to(dev)
    model.train()
    for critic in model.critics.values():  # no dropout anywhere: recorded and eager steps can be compared bit by bit
        for mod in critic.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    pool = [hv.make_batch(16, first_index=(rank * 4 + i) * 16).to(dev) for i in range(4)]
    ys = [b["y"] for b in pool]
    capacity = hv.GanTrainer.calibrate(model, pool).edge_capacity
    say("calibrated")
    # HIVT_REHEARSAL_OVERLAP=1: the chunked reductions that go out from grad-ready hooks during the backward pass
    # (over gloo nothing overlaps, but every hook, chunk and deferred weight-gradient launch runs as under RCCL)
    forced = os.environ.get("HIVT_REHEARSAL_OVERLAP") == "1"
    trainer = hv.GanTrainer(model, edge_capacity=capacity, overlap=True if forced else None)
    say("trainer built, distributed =", trainer.distributed, "overlap =", trainer.buckets["generator"].overlap)
    from hivt_gan_amd import ops as _ops
    grouped = {"launches": 0, "products": 0}
    plain_group = _ops._launch_wgrad_group

    def counting_group(flat, count, device):
        grouped["launches"] += 1
        grouped["products"] += count
        return plain_group(flat, count, device)
    _ops._launch_wgrad_group = counting_group

    def run_step(i):
        b = pool[i % len(pool)]
        b["y"] = ys[i % len(pool)]
        b["edge_plan"] = None
        return trainer.step(b)

    local = {}
    plain_sync = trainer._sync

    def recording_sync(group, bad=None):
        bucket = trainer.buckets[group]
        if group not in local:
            bucket.pack()
            local[group] = bucket.gradients()
        share = plain_sync(group, bad)
        local.setdefault(group + "_reduced", bucket.gradients())
        return share
    trainer._sync = recording_sync

    for i in range(steps):
        logs = run_step(i)
        torch.cuda.synchronize()
        say("step", i, {k: round(float(v), 5) for k, v in logs.items() if v.numel() == 1})
        state = torch.cat([p.detach().flatten() for p in model.parameters()])
        state = state if backend == "nccl" else state.cpu()
        both = [torch.empty_like(state) for _ in range(world)]
        dist.all_gather(both, state)
        assert torch.equal(both[0], both[1]), f"ranks diverged after step {i}"
        assert all(bool(torch.isfinite(v).all()) for v in logs.values())
        if i == 0 and not forced:  # (under the hooks the chunks are already reduced when the wrapper first looks)
            for group in ("critic", "generator"):
                mine = local[group] if backend == "nccl" else local[group].cpu()
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine)
                mean = ((parts[0] + parts[1]) / 2).cpu()
                err = float((local[group + "_reduced"].cpu() - mean).abs().max())
                scale = float(mean.abs().max())
                say(group, "reduced vs mean of locals: err", err, "scale", scale)
                assert scale > 0 and err <= 1e-6 * scale
    trainer._sync = plain_sync
    say("grouped weight-gradient launches", grouped)
    # weight gradients stay deferred and grouped under the hooks of the overlapped reduction too
    assert grouped["products"] > 20 * steps and grouped["launches"] < grouped["products"] // 4, grouped
    tuned = trainer.tune(run_step)
    say("tuned", tuned)
    for i in range(3):
        run_step(steps + i)
    torch.cuda.synchronize()
    # the same step recorded in graph segments with the reductions between them
    from hivt_gan_amd.graph import GraphedStep
    spec = hv.GanTrainer.calibrate(model, pool)
    caps = torch.tensor([spec.num_nodes, spec.num_lanes, spec.num_edges, spec.num_lane_actor, *spec.edge_capacity])
    caps = caps.to(dev) if backend == "nccl" else caps
    dist.all_reduce(caps, op=dist.ReduceOp.MAX)
    c = [int(v) for v in caps.tolist()]
    spec = hv.BatchSpec(c[0], c[1], c[2], c[3], edge_capacity=tuple(c[4:]))
    trainer.edge_capacity = spec.edge_capacity
    for b, y in zip(pool, ys):
        b["y"] = y
    padded = [hv.pad_batch(b, spec) for b in pool]
    recorded = GraphedStep(trainer, spec, padded[0])
    say("recorded", [k for k, _ in recorded.segments], recorded.nodes)
    for i in range(3):
        logs = recorded.run(padded[i % len(padded)])
        torch.cuda.synchronize()
        state = torch.cat([p.detach().flatten() for p in model.parameters()])
        state = state if backend == "nccl" else state.cpu()
        both = [torch.empty_like(state) for _ in range(world)]
        dist.all_gather(both, state)
        assert torch.equal(both[0], both[1]), f"ranks diverged after recorded step {i}"
        assert all(bool(torch.isfinite(v).all()) for v in logs.values())
        say("recorded step", i, round(float(logs["g_total"]), 5))
    # a recorded step and the eager step from the same state leave the same bits in every parameter (no dropout in
    # this configuration, so the two draw nothing differently)
    def parameters_now():
        return torch.cat([p.detach().flatten() for p in model.parameters()]).clone()

    snapshot = recorded._snapshot()
    recorded.run(padded[1])
    torch.cuda.synchronize()
    after_replay = parameters_now()
    recorded._restore(snapshot)
    eager = padded[1]
    eager["edge_plan"] = None
    trainer.step(eager)
    torch.cuda.synchronize()
    after_eager = parameters_now()
    differing = int((after_replay != after_eager).sum())
    say("recorded vs eager step: parameters that differ", differing, "of", after_eager.numel())
    assert differing == 0, "the recorded data-parallel step and the eager one disagree"
    dist.barrier()
    say("OK")
    faulthandler.cancel_dump_traceback_later()
    dist.destroy_process_group()


def main():
    os.makedirs(OUT, exist_ok=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port, 4), nprocs=2, join=True)
    print("dp rehearsal OK")


if __name__ == "__main__":
    main()
