"""Data-gradient and weight-gradient products: the library's kernels vs the stock GEMM (tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hivt_gan_amd import _lib, ops  # noqa: E402

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


for (M, K, N) in [(136_000, 128, 128), (136_000, 128, 256), (136_000, 256, 128), (69_000, 128, 128), (52_000, 128, 512),
                  (52_000, 512, 128), (52_000, 128, 128), (47_000, 128, 256), (15_738, 128, 128), (2_623, 256, 256)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    dy = torch.randn(M, N, device=dev)
    dx = torch.empty(M, K, device=dev)
    dw = torch.empty(N, K, device=dev)
    db = torch.empty(N, device=dev)
    ws = torch.empty(_lib.query("hivt_linear_wgrad_workspace_bytes", M, N, K), dtype=torch.uint8, device=dev)
    s = ops._stream(x)
    t_bd = timeit(lambda: _lib.call("hivt_linear_bwd_data", dy.data_ptr(), N, w.data_ptr(), K, M, N, K, None, 0, 1.0, None, 0,
                                    dx.data_ptr(), K, s))
    t_bd_s = timeit(lambda: torch.mm(dy, w, out=dx))
    t_wg = timeit(lambda: _lib.call("hivt_linear_wgrad", dy.data_ptr(), N, x.data_ptr(), K, M, N, K, dw.data_ptr(),
                                    db.data_ptr(), ws.data_ptr(), ws.numel(), s))
    t_wg_s = timeit(lambda: (torch.mm(dy.t(), x, out=dw), torch.sum(dy, 0, out=db)))
    t_wg_s2 = timeit(lambda: torch.mm(dy.t(), x, out=dw))
    fl = 2.0 * M * K * N / 1e6
    print(f"M={M:7d} K={K:4d} N={N:4d} | dX ours {t_bd:7.1f} us ({fl / t_bd:5.1f} TF) stock {t_bd_s:7.1f} us ({fl / t_bd_s:5.1f} TF)"
          f" | dW+db ours {t_wg:7.1f} us ({fl / t_wg:5.1f} TF) stock {t_wg_s:7.1f} us, dW only {t_wg_s2:7.1f} us ({fl / t_wg_s2:5.1f} TF)")
