"""A few launches of the linear kernels on the step's largest shapes (for counter passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hivt_gan_amd import ops  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
for (M, K, N, ln) in [(136_000, 128, 256, False), (136_000, 256, 128, True), (55_000, 128, 512, False),
                      (55_000, 512, 128, False), (52_000, 128, 128, True)]:
    x = torch.randn(M, K, device=dev, requires_grad=True)
    lin = torch.nn.Linear(K, N).to(dev)
    norm = torch.nn.LayerNorm(N).to(dev) if ln else None
    gy = torch.randn(M, N, device=dev)
    for _ in range(3):
        y = ops.fused_linear(x, lin.weight, lin.bias, layer_norm=norm, relu=ln)
        y.backward(gy)
torch.cuda.synchronize()
