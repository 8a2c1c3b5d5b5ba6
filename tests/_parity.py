"""Collects the error figures the GPU parity tests measure and writes them as one JSON file when the session ends
(``gpurun_out/parity_report.json`` under the repository root): per compared tensor the normwise relative error,
the worst elementwise relative error over the elements above 1e-3 of the tensor's largest (``worst_elementwise``), the bound it was held
to and -- for comparisons that allow twice the fp32 CPU reference's own error -- which branch of the bound passed."""
import json
import os

RECORDS = []


def errors(gpu, ref64):
    """(normwise relative error, worst elementwise relative error over elements above 1e-3 * max |ref|)."""
    g = gpu.detach().cpu().double().flatten()
    r = ref64.detach().cpu().double().flatten()
    if r.numel() == 0:
        return 0.0, 0.0
    norm = float(r.norm())
    normwise = float((g - r).norm()) / max(norm, 1e-300)
    big = r.abs() >= 1e-3 * float(r.abs().max())
    elementwise = float(((g - r).abs()[big] / r.abs()[big]).max()) if bool(big.any()) else 0.0
    return normwise, elementwise


def record(test: str, tensor: str, gpu, ref64, bound: float, branch: str = "rtol", **extra) -> float:
    normwise, elementwise = errors(gpu, ref64)
    RECORDS.append(dict(test=test, tensor=tensor, normwise=normwise, worst_elementwise=elementwise,
                        bound=bound, branch=branch, **extra))
    return normwise


def write(root: str) -> None:
    if not RECORDS:
        return
    out_dir = os.path.join(root, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    by_test = {}
    for r in RECORDS:
        by_test.setdefault(r["test"], []).append({k: v for k, v in r.items() if k != "test"})
    summary = {t: {"tensors": len(rs), "worst_normwise": max(r["normwise"] for r in rs),
                   "worst_elementwise": max(r["worst_elementwise"] for r in rs),
                   "passed_through_twice_the_fp32_reference_error": sum(1 for r in rs if r["branch"] == "2x fp32 reference")}
               for t, rs in by_test.items()}
    with open(os.path.join(out_dir, "parity_report.json"), "w") as f:
        json.dump({"summary": summary, "tensors": by_test}, f, indent=1)
