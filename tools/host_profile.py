"""Host-side cost of one training step: which framework ops the launch thread spends its time in."""
import os
import sys
import time

import torch
from torch.profiler import ProfilerActivity, profile

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


def step(i):
    b, y = pool[i % 2]
    b["y"] = y
    b["edge_plan"] = None
    return trainer.step(b)


for i in range(4):
    step(i)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for i in range(n):
    step(i)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"wall {t_all / n * 1e3:.2f} ms/step, host enqueue {t_enq / n * 1e3:.2f} ms/step")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
ev = prof.key_averages()
print(f"ops/step {sum(e.count for e in ev) / 3:.0f}, device time {sum(e.self_device_time_total for e in ev) / 3e3:.2f} ms/step")
print(ev.table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))

# Python-level view (backward moved onto the calling thread so that the profiler sees it)
import cProfile
import pstats

with torch.autograd.set_multithreading_enabled(False):
    pr = cProfile.Profile()
    pr.enable()
    for i in range(5):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
