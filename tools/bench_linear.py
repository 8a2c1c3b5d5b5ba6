"""Micro-benchmark: hand-written fp32-MFMA linear kernels vs the stock GPU ops, forward and backward."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hivt_gan_amd import ops

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
    return e0.elapsed_time(e1) / iters * 1e3  # us


for (M, K, N, ln, relu) in [(136_000, 128, 128, True, True), (136_000, 256, 128, True, True), (136_000, 128, 256, False, False),
                            (69_000, 128, 128, True, True), (52_000, 128, 512, False, True), (52_000, 512, 128, False, False),
                            (52_000, 128, 128, False, False), (2_600, 128, 128, False, False), (136_000, 2, 128, True, True)]:
    x = torch.randn(M, K, device=dev, requires_grad=True)
    lin = torch.nn.Linear(K, N).to(dev)
    norm = torch.nn.LayerNorm(N).to(dev) if ln else None
    gy = torch.randn(M, N, device=dev)

    def stock():
        y = lin(x)
        if ln:
            y = norm(y)
        if relu:
            y = torch.relu(y)
        return y

    def fused():
        return ops.fused_linear(x, lin.weight, lin.bias, layer_norm=norm, relu=relu)

    def fb(f):
        def g():
            y = f()
            y.backward(gy)
            x.grad = None
            lin.weight.grad = None
        return g

    ts, tf = timeit(stock), timeit(fused)
    tsb, tfb = timeit(fb(stock)), timeit(fb(fused))
    flops = 2.0 * M * K * N
    print(f"M={M:7d} K={K:4d} N={N:4d} ln={int(ln)} relu={int(relu)} | fwd stock {ts:8.1f} us fused {tf:8.1f} us "
          f"({flops / tf / 1e6:6.1f} TF) | fwd+bwd stock {tsb:8.1f} us fused {tfb:8.1f} us ({3 * flops / tfb / 1e6:6.1f} TF)")
