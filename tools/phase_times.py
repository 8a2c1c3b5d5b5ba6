"""GPU-side and host-side duration of the phases of a training step (tuning aid): events recorded on
the stream at phase boundaries next to host clock readings at the same points."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hivt_gan_amd as hv  # noqa: E402
from bench import model_config  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(2022)
model = hv.HiVT(**model_config(128)).to(dev)
model.train()
pool = [(b, b["y"]) for b in (hv.make_batch(128, first_index=i * 128).to(dev) for i in range(2))]
cap = hv.GanTrainer.calibrate(model, [b for b, _ in pool]).edge_capacity
trainer = hv.GanTrainer(model, edge_capacity=cap)

marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e, time.perf_counter()))


def wrap(obj, attr, name):
    fn = getattr(obj, attr)

    def inner(*a, **k):
        mark(name + ":begin")
        out = fn(*a, **k)
        mark(name + ":end")
        return out
    setattr(obj, attr, inner)


wrap(model, "forward", "generator forward")
wrap(model.d_loss_fn, "forward_masked", "critic loss forward")
wrap(model.g_loss_fn, "forward_masked", "generator adversarial forward")
wrap(trainer.opt_g, "step", "opt_g.step")
wrap(trainer.opt_d, "step", "opt_d.step")
orig_backward = torch.Tensor.backward


def backward(self, *a, **k):
    mark("backward:begin")
    out = orig_backward(self, *a, **k)
    mark("backward:end")
    return out


torch.Tensor.backward = backward


def step(i):
    b, y = pool[i % 2]
    b["y"] = y
    b["edge_plan"] = None
    mark("step:begin")
    out = trainer.step(b)
    mark("step:end")
    return out


for i in range(6):
    step(i)
torch.cuda.synchronize()
marks.clear()
n = 6
for i in range(n):
    step(i)
torch.cuda.synchronize()
# average over steps, by position inside a step
per_step = len(marks) // n
print(f"{'phase':45s} {'gpu ms':>8s} {'host ms':>8s} {'host lead ms':>12s}")
e0, h0 = marks[0][1], marks[0][2]
tot_g = tot_h = 0.0
for j in range(per_step - 1):
    g = h = lead = 0.0
    for s in range(n):
        a, b = marks[s * per_step + j], marks[s * per_step + j + 1]
        g += a[1].elapsed_time(b[1])
        h += (b[2] - a[2]) * 1e3
        # how far the host's enqueue point is ahead of the GPU's execution point at mark a
        lead += e0.elapsed_time(a[1]) - (a[2] - h0) * 1e3
    print(f"{marks[j][0] + ' -> ' + marks[j + 1][0]:45s} {g / n:8.2f} {h / n:8.2f} {lead / n:12.2f}")
    tot_g += g / n
    tot_h += h / n
print(f"{'sum inside a step':45s} {tot_g:8.2f} {tot_h:8.2f}")
print(f"whole loop: gpu {e0.elapsed_time(marks[-1][1]) / n:.2f} ms/step, host {(marks[-1][2] - h0) * 1e3 / n:.2f} ms/step")
