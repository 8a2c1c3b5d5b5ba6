"""Two data-parallel ranks sharing cuda:0 over gloo: the multi-rank control flow of ``GanTrainer`` with
the real GPU model (RCCL needs one device per rank, so this rehearses everything but the transport).

Each rank logs its stages to gpurun_out/dp_rank<r>.log and dumps its Python stack if it stalls.
Checks: ranks hold identical parameters after every step, the losses are finite, and the first
step's reduced generator gradient equals the mean of the two ranks' local gradients.
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
    pool = [hv.make_batch(16, first_index=(rank * 4 + i) * 16).to(dev) for i in range(4)]
    ys = [b["y"] for b in pool]
    capacity = hv.GanTrainer.calibrate(model, pool).edge_capacity
    say("calibrated")
    trainer = hv.GanTrainer(model, edge_capacity=capacity)
    say("trainer built, distributed =", trainer.distributed)

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
        if i == 0:
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
