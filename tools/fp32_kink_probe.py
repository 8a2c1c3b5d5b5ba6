"""How stable is the fp32 step's gradient error against the float64 oracle under changes that only move roundings?
Runs the precision report of tests/test_gpu_config3.py (HiVT-128, 8 scenes, fp32 mode) for several weight seeds and
scene sets and prints the whole-gradient error and the worst tensor of the smooth functional's gradient:
python tools/fp32_kink_probe.py [cases=6]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_config3 as T  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for i in range(cases):
    r = T.precision_report(modes=("fp32",), save=False, seed=2022 + i, first_index=100 * i)["fp32"]
    worst = max(r["smooth_grad_per_tensor"].items(), key=lambda kv: kv[1])
    over = sum(1 for v in r["smooth_grad_per_tensor"].values() if v > 1e-5)
    print(f"case {i}: smooth whole {r['smooth_grad_whole']:.2e} worst {worst[0]} {worst[1]:.2e} tensors over 1e-5: {over} "
          f"| loss whole {r['loss_grad_whole']:.2e}", flush=True)
