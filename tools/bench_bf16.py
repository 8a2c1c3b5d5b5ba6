"""fp32 vs bf16-operand mode of the dense products: microbenchmark of the three kernels (tuning aid)."""
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


for (M, K, N) in [(136_000, 128, 128), (136_000, 128, 256), (52_000, 128, 512), (52_000, 512, 128), (2_623, 256, 256),
                  (2_623, 128, 128), (2_623, 128, 768)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev)
    dx = torch.empty(M, K, device=dev)
    dw = torch.empty(N, K, device=dev)
    db = torch.empty(N, device=dev)
    ws = torch.empty(_lib.query("hivt_linear_wgrad_workspace_bytes", M, N, K), dtype=torch.uint8, device=dev)
    s = ops._stream(x)
    row = f"M={M:7d} K={K:4d} N={N:4d} |"
    ref = (x.double() @ w.double().t() + b.double())
    for mode in ("fp32", "bf16-plain", "bf16"):
        ops.set_matmul_precision(mode)
        tf = timeit(lambda: _lib.call("hivt_linear_fwd", x.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), None, 0, None, 0,
                                      None, None, 0.0, 0, 1.0, 0.0, 0, None, None, 0, M, N, K, y.data_ptr(), N, None, 0,
                                      None, s))
        tb = timeit(lambda: _lib.call("hivt_linear_bwd_data", dy.data_ptr(), N, w.data_ptr(), K, M, N, K, None, 0, 1.0,
                                      None, 0, dx.data_ptr(), K, s))
        tw = timeit(lambda: _lib.call("hivt_linear_wgrad", dy.data_ptr(), N, x.data_ptr(), K, M, N, K, dw.data_ptr(),
                                      db.data_ptr(), ws.data_ptr(), ws.numel(), s))
        err = float((y.double() - ref).norm() / ref.norm())
        row += f" {mode}: fwd {tf:6.1f} (err {err:.1e}) dX {tb:6.1f} dW {tw:6.1f} us |"
    ops.set_matmul_precision("fp32")
    print(row)
