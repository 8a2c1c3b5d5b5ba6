"""How the bf16 mode's worst per-tensor gradient error (tests/test_gpu_config3.py, bound 2e-2) spreads over model
seeds and batches, and what it becomes when one module's FORWARD products run on fp32 operands instead of split
bf16 (its backward products stay bf16): python tools/precision_spread.py [module ...], modules by attribute path under
the model, e.g. local_encoder.aa_encoder."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hivt_gan_amd as hv  # noqa: E402
from hivt_gan_amd import ops  # noqa: E402
from tests.test_gpu_config3 import precision_report  # noqa: E402

paths = sys.argv[1:]
original_init = hv.HiVT.__init__


def patched_init(self, *args, **kwargs):
    original_init(self, *args, **kwargs)
    for path in paths:
        mod = self
        for part in path.split("."):
            mod = getattr(mod, part)
        inner = mod.forward

        def forward(*a, _inner=inner, **k):
            prev = ops.set_matmul_precision("bf16-bwd")  # fp32 forward operands, bf16 gradient products
            try:
                return _inner(*a, **k)
            finally:
                ops.set_matmul_precision(prev)
        mod.forward = forward


hv.HiVT.__init__ = patched_init
print("forward on fp32 operands in:", paths or "nothing")
for seed, first in ((2022, 0), (2022, 8), (1, 0), (2, 16), (3, 24)):
    r = precision_report(modes=("bf16",), save=False, seed=seed, first_index=first)["bf16"]
    worst = sorted(r["loss_grad_per_tensor"].items(), key=lambda kv: -kv[1])[:3]
    smooth = sorted(r["smooth_grad_per_tensor"].items(), key=lambda kv: -kv[1])[:1]
    print(f"seed {seed} batch {first}: loss grad whole {r['loss_grad_whole']:.4f} worst "
          + ", ".join(f"{k.replace('local_encoder.', '').replace('global_interactor.', 'gi.')} {v:.4f}" for k, v in worst)
          + f" | smooth worst {smooth[0][0].replace('local_encoder.', '')} {smooth[0][1]:.4f}", flush=True)
