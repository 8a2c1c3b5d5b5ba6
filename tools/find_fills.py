"""Where the step's fill / zero / copy launches come from: a dispatch mode counts every aten call of
the watched kinds in one training step, keyed by operator, shape and the nearest frame of this
package (or "autograd engine" when the call has no Python frame of ours).

    python tools/find_fills.py [fill|copy|add|cat|all]
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hivt_gan_amd as hv  # noqa: E402
from bench import model_config  # noqa: E402

KINDS = {
    "fill": ("fill", "zero", "ones", "full"),
    "copy": ("copy", "clone", "contiguous", "_to_copy"),
    "add": ("add", "mul", "sub", "div", "where", "sum", "mean"),
    "cat": ("cat", "stack", "index", "gather", "scatter"),
}


class Watch(TorchDispatchMode):
    def __init__(self, needles):
        super().__init__()
        self.needles = needles
        self.seen = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if self.needles is None or any(n in name for n in self.needles):
            where = "autograd engine"
            for fr in reversed(traceback.extract_stack()):
                if "hivt_gan_amd" in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            shape = tuple(out.shape) if isinstance(out, torch.Tensor) else None
            dev = out.device.type if isinstance(out, torch.Tensor) else "-"
            self.seen[(name, shape, dev, where)] += 1
        return out


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "fill"
    needles = None if kind == "all" else KINDS[kind]
    dev = torch.device("cuda", 0)
    torch.manual_seed(2022)
    model = hv.HiVT(**model_config(128)).to(dev)
    model.train()
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
    with Watch(needles) as w:
        step(3)
    torch.cuda.synchronize()
    total = 0
    for (name, shape, d, where), n in sorted(w.seen.items(), key=lambda kv: -kv[1]):
        print(f"{n:4d}  {name:28s} {str(shape):22s} {d:5s} {where}")
        total += n
    print("total", total)


if __name__ == "__main__":
    main()
