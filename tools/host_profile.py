"""cProfile of the host side of the training step (where the enqueue time goes, by function)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hivt_gan_amd as hv  # noqa: E402
from bench import model_config  # noqa: E402

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


for i in range(5):
    step(i)
torch.cuda.synchronize()
N = 20
prof = cProfile.Profile()
prof.enable()
for i in range(N):
    step(i)
prof.disable()
torch.cuda.synchronize()
st = pstats.Stats(prof)
st.sort_stats("tottime")
print(f"per step: total {st.total_tt / N * 1e3:.2f} ms over {N} steps (profiler overhead included)")
st.print_stats(45)
st.sort_stats("cumulative")
st.print_stats(60)
