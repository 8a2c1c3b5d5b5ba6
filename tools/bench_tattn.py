"""Temporal attention core alone: python tools/bench_tattn.py [N=2752] -- microseconds per forward / backward
launch at S = 21, D = 128, 8 heads, dropout 0.1, in both matmul modes (bf16: matrix-core kernels, fp32: scalar)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hivt_gan_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2752
dev = torch.device("cuda", 0)
S, D, H = 21, 128, 8
for mode in ("bf16", "fp32"):
    ops.set_matmul_precision(mode)
    qkv = torch.randn(S, N, 3 * D, device=dev, requires_grad=True)
    w = torch.randn(S, N, D, device=dev)
    for name in ("fwd", "fwd+bwd"):
        times = []
        for it in range(25):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            qkv.grad = None
            e0.record()
            o = ops.temporal_attention(qkv, H, dropout_p=0.1, training=True)
            if name != "fwd":
                o.backward(w)
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
        times = sorted(times[5:])
        print(f"{mode} {name}: median {times[len(times) // 2]:.1f} us (min {times[0]:.1f})")
