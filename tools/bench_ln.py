"""LayerNorm forward / backward kernels against their HBM byte counts (tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hivt_gan_amd import _lib, ops  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=30, warm=5):
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


for M in (2615, 15738, 52305, 136000):
    x = torch.randn(M, 128, device=dev)
    g = torch.randn(128, device=dev)
    b = torch.randn(128, device=dev)
    dy = torch.randn(M, 128, device=dev)
    y = torch.empty_like(x)
    dz = torch.empty_like(x)
    stats = torch.empty(M, 2, device=dev)
    dg, db = torch.empty_like(g), torch.empty_like(b)
    ws = torch.empty(_lib.query("hivt_ln_bwd_workspace_bytes", M, 128), dtype=torch.uint8, device=dev)
    s = ops._stream(x)
    t_f = timeit(lambda: _lib.call("hivt_ln_fwd", x.data_ptr(), 128, g.data_ptr(), b.data_ptr(), 1e-5, M, 128,
                                   y.data_ptr(), 128, stats.data_ptr(), s))
    t_b = timeit(lambda: _lib.call("hivt_ln_bwd", dy.data_ptr(), 128, x.data_ptr(), 128, stats.data_ptr(), g.data_ptr(),
                                   b.data_ptr(), 0, M, 128, None, 0, dz.data_ptr(), 128, dg.data_ptr(), db.data_ptr(),
                                   ws.data_ptr(), ws.numel(), s))
    by = M * 128 * 4
    print(f"M={M:7d}  fwd {t_f:6.1f} us ({2 * by / t_f / 1e6:5.2f} TB/s)   bwd (kernel + reduce) {t_b:6.1f} us "
          f"({3 * by / t_b / 1e6:5.2f} TB/s)")
