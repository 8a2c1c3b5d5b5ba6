"""How much of a training step is host enqueue time vs device time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hivt_gan_amd as hv
from bench import model_config

dev = torch.device("cuda:0")
torch.manual_seed(2022)
for guards in (True, False):
    model = hv.HiVT(**model_config(128), check_finite=guards).to(dev)
    model.train()
    trainer = hv.GanTrainer(model)
    pool = [(b, b["y"]) for b in (hv.make_batch(128, first_index=i * 128).to(dev) for i in range(2))]

    def step(i):
        b, y = pool[i % 2]
        b["y"] = y
        b["edge_plan"] = None
        return trainer.step(b)

    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    n = 10
    host = 0.0
    t0 = time.perf_counter()
    for i in range(n):
        h0 = time.perf_counter()
        step(i)
        host += time.perf_counter() - h0
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # device-only time of the same steps: events around the loop measure GPU busy + idle; use a
    # profile of kernel time instead: run once more under the autograd profiler
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
    ev = prof.key_averages()
    dev_us = sum(e.device_time_total for e in ev if getattr(e, "device_type", None) is not None and e.device_time_total > 0 and e.self_device_time_total > 0)
    self_dev = sum(e.self_device_time_total for e in ev)
    print(f"guards={guards}: wall {t_all / n * 1e3:.1f} ms/step, host-side return {t_enq / n * 1e3:.1f} ms/step, "
          f"device kernel time {self_dev / 3 / 1e3:.1f} ms/step, ops/step {sum(e.count for e in ev) / 3:.0f}")
