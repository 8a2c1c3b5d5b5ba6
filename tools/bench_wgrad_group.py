"""Grouped weight gradients at the training step's shapes (tuning aid): one call with the four products of a
temporal-encoder layer (55 000 rows) and one with the per-edge products of the lane stage (136 000 rows)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hivt_gan_amd import ops  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def terms(m, shapes):
    out = []
    for n, k in shapes:
        out.append((torch.randn(m, n, device=dev), torch.randn(m, k, device=dev), torch.empty(n, k, device=dev),
                    torch.empty(n, device=dev), False, False))
    return out


for mode in ("fp32", "bf16"):
    ops.set_matmul_precision(mode)
    for name, m, shapes in (("temporal layer", 55_000, [(384, 128), (128, 128), (512, 128), (128, 512)]),
                            ("lane edges", 136_000, [(128, 256), (128, 128), (256, 128)]),
                            ("agent nodes", 30_000, [(128, 128), (128, 128), (256, 128), (128, 128), (512, 128), (128, 512)])):
        t = terms(m, shapes)
        us = timeit(lambda: ops.linear_wgrad_group(t))
        gb = sum(m * (n + k) * 4 for n, k in shapes) / 1e9
        print(f"{mode:5s} {name:15s} {us:7.1f} us  {gb / us * 1e6 / 1e3:5.2f} TB/s of operand bytes")
ops.set_matmul_precision("fp32")
