"""What plain streaming kernels of the framework take for launches of the edge-attention kernels' sizes
(context for their roofline fractions): a copy (1 read + 1 write) and an add (2 reads + 1 write) whose total
traffic equals the algorithmic bytes of each attention launch kind, timed per launch with events, rotating over
enough buffer sets that no launch finds its operands in the 256 MB last-level cache.

    python tools/hbm_rate_sized.py [--mb 54 104 127 143 247 285]
"""
import argparse
import statistics

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=float, nargs="*", default=[54.0, 104.0, 127.0, 143.0, 247.0, 285.0])
ap.add_argument("--reps", type=int, default=40)
args = ap.parse_args()
dev = "cuda:0"
for mb in args.mb:
    total = int(mb * 1e6)
    for name, streams in (("copy (1 read + 1 write)", 2), ("add (2 reads + 1 write)", 3)):
        n = total // (4 * streams)
        sets = max(2, int(600e6 // (4 * n * streams)) + 1)  # more than twice the last-level cache in rotation
        bufs = [[torch.randn(n, device=dev) for _ in range(streams)] for _ in range(sets)]
        times = []
        for rep in range(args.reps + 5):
            b = bufs[rep % sets]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if streams == 2:
                b[1].copy_(b[0])
            else:
                torch.add(b[0], b[1], out=b[2])
            e1.record()
            torch.cuda.synchronize()
            if rep >= 5:
                times.append(e0.elapsed_time(e1) * 1e3)
        us = statistics.median(times)
        print(f"{mb:6.0f} MB  {name:24s} median {us:6.1f} us  = {total / us / 1e6:5.2f} TB/s = {total / us / 1e6 / 8:.3f} of 8 TB/s")
        del bufs
