"""Host cost of one training step: the step on a batch so small that the GPU never limits it (same
launch count and Python work as the full workload), in both precisions."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hivt_gan_amd as hv  # noqa: E402
from hivt_gan_amd import ops  # noqa: E402
from bench import model_config  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(2022)
model = hv.HiVT(**model_config(128)).to(dev)
model.train()
pool = [hv.make_batch(8, first_index=i * 8).to(dev) for i in range(4)]
trainer = hv.GanTrainer(model, edge_capacity=hv.GanTrainer.calibrate(model, pool).edge_capacity)
ys = [b["y"] for b in pool]


def step(i):
    b = pool[i % 4]
    b["y"] = ys[i % 4]
    b["edge_plan"] = None
    return trainer.step(b)


for mode in ("fp32", "bf16"):
    ops.set_matmul_precision(mode)
    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    print(f"{mode}: {best * 1e3:.2f} ms/step on 8 scenes (host-bound)")
