"""Which gradient tensors the backward products of one training step read, and how large they are: every
data-gradient call and every (deferred) weight-gradient term of one eager step is listed with its caller, its
shape and the bytes of its dy operand, fp32 and bfloat16 ones apart -- the list of candidates for the bfloat16
memory form (a dy that only bf16 products read), largest first.

    python tools/grad_traffic.py
"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hivt_gan_amd as hv  # noqa: E402
from hivt_gan_amd import _lib, ops  # noqa: E402
from bench import model_config  # noqa: E402

seen = collections.Counter()
count = collections.Counter()


def where():
    for fr in reversed(traceback.extract_stack()):
        if fr.filename.endswith("ops.py") and fr.name in ("backward", "forward"):
            return f"ops.py:{fr.lineno}"
    return "?"


def note(kind, m, n, k, half):
    key = (kind, where(), m, n, k, "bf16" if half else "fp32")
    seen[key] += m * n * (2 if half else 4)
    count[key] += 1


real_call = _lib.call


def call(name, *args):
    if name in ("hivt_linear_bwd_data", "hivt_linear_bwd_data_half"):
        half = name.endswith("_half") and (args[14] & 1)
        note("dgrad", args[4], args[5], args[6], half)
    elif name in ("hivt_linear_wgrad", "hivt_linear_wgrad_half"):
        note("wgrad", args[4], args[5], args[6], name.endswith("_half"))
    return real_call(name, *args)


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(2022)
    model = hv.HiVT(**model_config(128)).to(dev)
    model.train()
    ops.set_matmul_precision("bf16")
    pool = [hv.make_batch(128, first_index=i * 128).to(dev) for i in range(2)]
    capacity = hv.GanTrainer.calibrate(model, pool).edge_capacity
    trainer = hv.GanTrainer(model, edge_capacity=capacity)
    ys = [b["y"] for b in pool]

    def step(i):
        b = pool[i % 2]
        b["y"] = ys[i % 2]
        b["edge_plan"] = None
        return trainer.step(b)

    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    _lib.call = call
    real_weight_grad = ops._weight_grad

    def weight_grad(g, g_ld, x, x_ld, m, n, k, dev_, bias=True, leaves=False):
        if leaves and ops._deferred is not None and m > 0:
            note("wgrad", m, n, k, g.dtype == torch.bfloat16)
        return real_weight_grad(g, g_ld, x, x_ld, m, n, k, dev_, bias=bias, leaves=leaves)

    ops._weight_grad = weight_grad
    step(2)
    torch.cuda.synchronize()
    total = collections.Counter()
    for key, nbytes in sorted(seen.items(), key=lambda kv: -kv[1]):
        kind, site, m, n, k, form = key
        total[form] += nbytes
        if nbytes >= 4 << 20:
            print(f"{nbytes / 1e6:8.1f} MB  {count[key]:3d}x  {kind:5s} {form}  dy [{m}, {n}] K={k}  {site}")
    print({f: f"{b / 1e6:.0f} MB" for f, b in total.items()})


if __name__ == "__main__":
    main()
