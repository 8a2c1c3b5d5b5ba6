"""The CPU side of tools/fp32_kink_probe.py: the oracle itself run in float32 against the oracle in float64 on the
same cases (HiVT-128, 8 scenes): whole-gradient error of the smooth functional and of the supervised loss.  Shows
what an fp32 evaluation with the framework's own operators is from float64 on these cases.
python tools/fp32_kink_probe_cpu.py [cases=6]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hivt_gan_amd as hv  # noqa: E402
from oracle import hivt_oracle as O  # noqa: E402

CFG128 = dict(historical_steps=20, future_steps=30, num_modes=6, rotate=True, node_dim=2, edge_dim=2, embed_dim=128,
              num_heads=8, dropout=0.0, num_temporal_layers=4, num_global_layers=3, local_radius=50.0, parallel=False)
SHIFT_INVARIANT = ("lin_k.bias", "lin_k_node.bias", "lin_k_edge.bias", "decoder.pi.6.bias")


def grads(model, b, dtype):
    names = [k for k, _ in model.named_parameters()]
    sd = {k: (v.detach().to(dtype) if v.dtype.is_floating_point else v.detach().clone()) for k, v in model.state_dict().items()}
    for k in names:
        sd[k].requires_grad_(True)
    data = {k: (b[k].to(dtype) if torch.is_tensor(b[k]) and b[k].dtype == torch.float32 else b[k]) for k in b.keys()}
    y, pi, y_rot = O.hivt_forward(sd, data, dict(CFG128), training=False)
    gen = torch.Generator().manual_seed(3)
    wy = torch.randn(y.shape, generator=gen).to(dtype)
    wp = torch.randn(pi.shape, generator=gen).to(dtype)
    params = [sd[k] for k in names]
    smooth = torch.autograd.grad((y * wy).sum() + (pi * wp).sum(), params, retain_graph=True, allow_unused=True)
    reg, cls, best = O.supervised_losses(y, pi, y_rot, b["padding_mask"], 20)
    loss = torch.autograd.grad(reg + cls, params, allow_unused=True)
    return names, smooth, loss, best


def whole(names, got, want):
    num = den = 0.0
    worst = ("", 0.0)
    for k, g, w in zip(names, got, want):
        if w is None or k.endswith(SHIFT_INVARIANT) or float(w.norm()) == 0.0:
            continue
        e = float((g.double() - w).norm() / w.norm())
        if e > worst[1]:
            worst = (k, e)
        num += float((g.double() - w).pow(2).sum())
        den += float(w.pow(2).sum())
    return (num / den) ** 0.5, worst


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for i in range(cases):
    torch.manual_seed(2022 + i)
    model = hv.HiVT(**CFG128, lr=5e-4, weight_decay=1e-4, T_max=64)
    b = hv.make_batch(8, first_index=100 * i)
    names, s64, l64, best64 = grads(model, b, torch.float64)
    _, s32, l32, best32 = grads(model, b, torch.float32)
    ws, worst = whole(names, s32, s64)
    wl, _ = whole(names, l32, l64)
    print(f"case {i}: fp32 CPU oracle vs float64: smooth whole {ws:.2e} worst {worst[0]} {worst[1]:.2e} | loss whole {wl:.2e} "
          f"| best modes differ for {int((best32 != best64).sum())} agents", flush=True)
