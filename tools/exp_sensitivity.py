"""Which forward tensors may be kept in bfloat16 without pushing the parameter gradients past 2e-2?
Runs the HiVT-128 precision report (tests/test_gpu_config3.py) in fp32 mode with individual forward
tensors rounded to bfloat16 (straight-through: the backward sees the identity)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from hivt_gan_amd import ops  # noqa: E402
import test_gpu_config3 as T  # noqa: E402


class RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def run(name):
    r = T.precision_report(modes=("fp32",), save=False)["fp32"]
    print(f"## {name}: y_hat {r['y_hat']:.2e} smooth whole {r['smooth_grad_whole']:.2e} worst {r['smooth_grad_worst'][1]:.2e} "
          f"| loss whole {r['loss_grad_whole']:.2e} worst {r['loss_grad_worst'][1]:.2e}", flush=True)


orig_seg = ops.seg_attn
orig_fused = ops.fused_linear
orig_pair = ops.pair_linear

run("baseline")
ops.seg_attn = lambda q, kv, *a, **k: orig_seg(q, RoundSTE.apply(kv), *a, **k)
run("K|V rounded")
ops.seg_attn = lambda q, kv, *a, **k: orig_seg(RoundSTE.apply(q), RoundSTE.apply(kv), *a, **k)
run("Q and K|V rounded")
ops.seg_attn = orig_seg
ops.fused_linear = lambda *a, **k: RoundSTE.apply(orig_fused(*a, **k))
run("every fused_linear output rounded")
ops.fused_linear = lambda x, *a, **k: orig_fused(RoundSTE.apply(x), *a, **k)
run("every fused_linear input rounded")
ops.fused_linear = orig_fused
