"""Which framework operators of one training step issue a runtime memset (hipMemsetAsync)?  A recorded HIP
graph must hold none: on this stack (torch 2.10 / ROCm 7.2) a memset node takes no effect when the graph is
replayed (tools/graph_probe.py --stage tiny shows it on a reduction's semaphore buffer).

One eager read-back-free step runs under the profiler; every runtime memset call is attributed to the
innermost operator whose time range contains it and to the nearest Python frame of this package.
"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hivt_gan_amd as hv  # noqa: E402
from bench import model_config  # noqa: E402


def main():
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda", 0)
    torch.manual_seed(2022)
    model = hv.HiVT(**model_config(128)).to(dev)
    model.train()
    pool = [hv.make_batch(scenes, first_index=i * scenes).to(dev) for i in range(2)]
    spec = hv.GanTrainer.calibrate(model, pool)
    pool = [hv.pad_batch(b, spec) for b in pool]
    trainer = hv.GanTrainer(model, edge_capacity=spec.edge_capacity)
    from hivt_gan_amd import losses
    losses.CONCURRENT_CRITICS = False
    ys = [b["y"] for b in pool]

    def step(i):
        b = pool[i % 2]
        b["y"] = ys[i % 2]
        b["edge_plan"] = None
        return trainer.step(b)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step(3)
        torch.cuda.synchronize()
    events = list(prof.events())
    ops_ = [e for e in events if e.device_type == torch.autograd.DeviceType.CPU and not e.name.startswith("hip")]
    hits = [e for e in events if "memset" in e.name.lower() and e.device_type == torch.autograd.DeviceType.CPU]
    seen = collections.Counter()
    for h in hits:
        t = h.time_range.start
        inner = None
        for o in ops_:
            if o.time_range.start <= t <= o.time_range.end:
                if inner is None or (o.time_range.end - o.time_range.start) < (inner.time_range.end - inner.time_range.start):
                    inner = o
        where = "?"
        name = inner.name if inner is not None else "?"
        # the stack of the innermost op, or of an enclosing one that carries a frame of this package
        cands = sorted([o for o in ops_ if o.time_range.start <= t <= o.time_range.end],
                       key=lambda o: o.time_range.end - o.time_range.start)
        for o in cands:
            frames = [f for f in (o.stack or []) if "hivt_gan_amd" in f or "bench.py" in f]
            if frames:
                where = frames[0].strip()
                break
        shapes = str(inner.input_shapes) if inner is not None and inner.input_shapes else ""
        outer = [o.name for o in cands[1:4]]
        shapes += f"  inside {outer}"
        seen[(name, where, shapes)] += 1
    print(f"{len(hits)} runtime memset calls in one step ({scenes} scenes)")
    for (name, where, shapes), n in sorted(seen.items(), key=lambda kv: -kv[1]):
        print(f"{n:3d}  {name:40s} {where}  {shapes}")
    names = collections.Counter(e.name for e in events if e.device_type == torch.autograd.DeviceType.CPU
                                and e.name.startswith("hip"))
    print("runtime calls:", dict(names))


if __name__ == "__main__":
    main()
