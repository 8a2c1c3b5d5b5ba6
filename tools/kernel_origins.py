"""Which operator launches the step's small framework kernels: one eager training step under the framework's
profiler, every device kernel whose name contains one of the needles is attributed to the CPU operator that
launched it and to that operator's chain of parents (up to the autograd node or the Python-side record).

    python tools/kernel_origins.py [needle ...]      (default: FillFunctor CUDAFunctor_add copy)
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
    needles = sys.argv[1:] or ["FillFunctor", "CUDAFunctor_add", "copy"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(2022)
    model = hv.HiVT(**model_config(128)).to(dev)
    model.train()
    hv.ops.set_matmul_precision("bf16")
    pool = [hv.make_batch(128, first_index=i * 128).to(dev) for i in range(2)]
    capacity = hv.GanTrainer.calibrate(model, pool).edge_capacity
    trainer = hv.GanTrainer(model, edge_capacity=capacity)
    ys = [b["y"] for b in pool]

    def step(i):
        b = pool[i % 2]
        b["y"] = ys[i % 2]
        b["edge_plan"] = None
        return trainer.step(b)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step(3)
        torch.cuda.synchronize()
    seen = collections.Counter()
    micros = collections.Counter()
    for ev in prof.events():
        for k in getattr(ev, "kernels", []):
            if not any(n in k.name for n in needles):
                continue
            chain, p = [], ev
            while p is not None and len(chain) < 6:
                chain.append(p.name)
                p = p.cpu_parent
            shapes = str(ev.input_shapes)[:60] if ev.input_shapes else ""
            short = next(n for n in needles if n in k.name)
            key = (short, " <- ".join(chain), shapes)
            seen[key] += 1
            micros[key] += k.duration
    total = 0
    for key, n in sorted(seen.items(), key=lambda kv: -kv[1]):
        print(f"{n:4d} {micros[key]:8.1f}us  {key[0]:16s} {key[1]}   {key[2]}")
        total += n
    print("total", total)


if __name__ == "__main__":
    main()
