"""A few launches of the linear kernels on the step's small shapes (for kernel traces)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hivt_gan_amd import ops  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
for (M, K, N) in [(2623, 128, 128), (2623, 256, 256), (2623, 128, 512), (2623, 512, 128), (15738, 256, 128)]:
    x = torch.randn(M, K, device=dev, requires_grad=True)
    lin = torch.nn.Linear(K, N).to(dev)
    gy = torch.randn(M, N, device=dev)
    for _ in range(5):
        y = ops.fused_linear(x, lin.weight, lin.bias)
        y.backward(gy)
        torch.cuda.synchronize()
torch.cuda.synchronize()
