"""The named bf16 workload (HiVT-128, dense products on bf16 operands) against the CPU oracle run in
float64 on the same weights and the same synthetic scenes: forward outputs, supervised losses and every
parameter gradient, per tensor.  The fp32 mode goes through the same comparison, which gives every
bound below a measured fp32 figure to stand next to.

Two gradients are compared.  (1) The gradient of a fixed linear functional of the outputs: nothing
but the network's backward pass, so its error is the products' rounding and nothing else.  (2) The
gradient of the supervised loss, which picks the best of six modes per agent; an agent whose two best
modes are within rounding of each other can change winner under bf16, and the test therefore feeds
both sides the float64 oracle's winners when it compares this gradient (the selection itself is
compared separately)."""
import json
import os

import pytest
import torch

import hivt_gan_amd as hv
from hivt_gan_amd import ops
from oracle import hivt_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)

CFG128 = dict(historical_steps=20, future_steps=30, num_modes=6, rotate=True, node_dim=2, edge_dim=2, embed_dim=128,
              num_heads=8, dropout=0.0, num_temporal_layers=4, num_global_layers=3, local_radius=50.0, parallel=False)
SHIFT_INVARIANT = ("lin_k.bias", "lin_k_node.bias", "lin_k_edge.bias", "decoder.pi.6.bias")
SCENES = 8


def _rel(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    return float((a - b).norm() / b.norm().clamp(min=1e-300))


def _loss_with_given_best(y_hat, pi, y, padding_mask, best, t_hist=20):
    """models/hivt.py:186-209 with the winning mode supplied instead of taken by argmin."""
    reg_mask = ~padding_mask[:, t_hist:]
    valid_steps = reg_mask.sum(dim=-1)
    cls_mask = valid_steps > 0
    l2 = (torch.norm(y_hat[..., :2] - y, p=2, dim=-1) * reg_mask).sum(dim=-1)
    y_best = y_hat[best, torch.arange(y.shape[0], device=y.device)]
    reg = O.laplace_nll(y_best[reg_mask], y[reg_mask])
    soft = torch.softmax(-l2[:, cls_mask] / (valid_steps[cls_mask] + 1e-6), dim=0).t().detach()
    cls = O.soft_target_cross_entropy(pi[cls_mask], soft)
    return reg, cls


def precision_report(modes=("fp32", "bf16", "bf16-plain", "bf16-bwd", "bf16-split-fwd"), save=True, seed=2022,
                     first_index=0):
    """Per-mode relative errors of outputs, losses and every parameter gradient against the float64 oracle."""
    torch.manual_seed(seed)
    model = hv.HiVT(**CFG128, lr=5e-4, weight_decay=1e-4, T_max=64).to(DEV)
    model.train()
    b = hv.make_batch(SCENES, first_index=first_index)
    names = [k for k, _ in model.named_parameters()]
    # float64 oracle
    sd = {k: (v.detach().cpu().double() if v.dtype.is_floating_point else v.detach().cpu().clone())
          for k, v in model.state_dict().items()}
    for k in names:
        sd[k].requires_grad_(True)
    data64 = {k: (b[k].double() if torch.is_tensor(b[k]) and b[k].dtype == torch.float32 else b[k]) for k in b.keys()}
    y_ref, pi_ref, y_rot = O.hivt_forward(sd, data64, dict(CFG128), training=False)
    gen = torch.Generator().manual_seed(3)
    wy = torch.randn(y_ref.shape, generator=gen)
    wp = torch.randn(pi_ref.shape, generator=gen)
    params64 = [sd[k] for k in names]
    smooth_ref = torch.autograd.grad((y_ref * wy.double()).sum() + (pi_ref * wp.double()).sum(), params64,
                                     retain_graph=True, allow_unused=True)
    reg_ref, cls_ref, best_ref = O.supervised_losses(y_ref, pi_ref, y_rot, b["padding_mask"], 20)
    loss_ref = torch.autograd.grad(reg_ref + cls_ref, params64, allow_unused=True)
    ref = dict(y_hat=y_ref.detach(), pi=pi_ref.detach(), reg=reg_ref.detach(), cls=cls_ref.detach(), best=best_ref,
               smooth=dict(zip(names, smooth_ref)), loss=dict(zip(names, loss_ref)))

    out = {}
    for mode in modes:
        prev = ops.set_matmul_precision(mode)
        try:
            res = {}
            g = b.to(DEV)
            y_hat, pi = model(g)
            params = [p for _, p in model.named_parameters()]
            sm = torch.autograd.grad((y_hat * wy.to(DEV)).sum() + (pi * wp.to(DEV)).sum(), params, retain_graph=True,
                                     allow_unused=True)
            reg_own, cls_own, best_own = model._supervised_losses(g, y_hat, pi)
            reg, cls = _loss_with_given_best(y_hat, pi, g["y"], g["padding_mask"], best_ref.to(DEV))
            lg = torch.autograd.grad(reg + cls, params, allow_unused=True)
            valid = (~b["padding_mask"][:, 20:]).any(dim=1)
            res["y_hat"] = _rel(y_hat, ref["y_hat"])
            res["pi"] = _rel(pi, ref["pi"])
            res["reg_loss"] = abs(float(reg_own) - float(ref["reg"])) / abs(float(ref["reg"]))
            res["cls_loss"] = abs(float(cls_own) - float(ref["cls"])) / abs(float(ref["cls"]))
            res["best_mode_agreement"] = float((best_own.cpu()[valid] == ref["best"][valid]).double().mean())
            for kind, got, want in (("smooth", sm, ref["smooth"]), ("loss", lg, ref["loss"])):
                per = {}
                num = den = 0.0
                for k, gt in zip(names, got):
                    w = want[k]
                    if w is None or k.endswith(SHIFT_INVARIANT) or float(w.norm()) == 0.0:
                        continue
                    per[k] = _rel(gt, w)
                    num += float((gt.detach().cpu().double() - w).pow(2).sum())
                    den += float(w.pow(2).sum())
                res[f"{kind}_grad_per_tensor"] = per
                res[f"{kind}_grad_whole"] = (num / den) ** 0.5
                res[f"{kind}_grad_worst"] = max(per.items(), key=lambda kv: kv[1])
            out[mode] = res
        finally:
            ops.set_matmul_precision(prev)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        if not save:
            raise OSError
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config3_precision_report.json"), "w") as f:
            json.dump(out, f, indent=1)
    except OSError:
        pass
    for mode, res in out.items():
        print(f"[{mode}] y_hat {res['y_hat']:.2e} pi {res['pi']:.2e} reg {res['reg_loss']:.2e} cls {res['cls_loss']:.2e} "
              f"best-mode agreement {res['best_mode_agreement']:.4f} | smooth grad whole {res['smooth_grad_whole']:.2e} "
              f"worst {res['smooth_grad_worst'][0]} {res['smooth_grad_worst'][1]:.2e} | loss grad whole "
              f"{res['loss_grad_whole']:.2e} worst {res['loss_grad_worst'][0]} {res['loss_grad_worst'][1]:.2e}")
    return out


@pytest.fixture(scope="module")
def report():
    return precision_report()


def _worst(per):
    return max(per.items(), key=lambda kv: kv[1])


def test_fp32_mode_hivt128_within_1e5_of_the_float64_oracle(report):
    r = report["fp32"]
    assert r["y_hat"] < 1e-5 and r["pi"] < 1e-5 and r["reg_loss"] < 1e-5 and r["cls_loss"] < 1e-5, r
    assert r["best_mode_agreement"] == 1.0
    for kind in ("smooth", "loss"):
        assert r[f"{kind}_grad_whole"] < 1e-5
        name, err = _worst(r[f"{kind}_grad_per_tensor"])
        assert err < 1e-5, f"{kind} gradient of {name}: {err:.2e}"


def test_bf16_mode_hivt128_within_2e2_of_the_float64_oracle(report):
    """The named workload's precision: forward products on split bf16 operands, gradient products on bf16
    operands.  Outputs, losses and EVERY parameter gradient stay within 2e-2 (relative L2 per tensor)."""
    r = report["bf16"]
    assert 1e-6 < r["y_hat"] < 1e-3 and r["pi"] < 1e-3, (r["y_hat"], r["pi"])
    assert r["reg_loss"] < 1e-4 and r["cls_loss"] < 1e-4
    assert r["best_mode_agreement"] == 1.0
    for kind in ("smooth", "loss"):
        assert r[f"{kind}_grad_whole"] < 1e-2, (kind, r[f"{kind}_grad_whole"])
        name, err = _worst(r[f"{kind}_grad_per_tensor"])
        assert err < 2e-2, f"{kind} gradient of {name}: {err:.2e}"


def test_plain_bf16_forward_operands_do_not_hold_2e2_on_the_gradients(report):
    """Why the forward product is split: with its operands rounded to 8 significant bits the outputs stay
    inside 2e-2 but the parameter gradients, which depend on the saved activations, move by about 1e-1.
    (Rounding only the gradient products keeps every tensor within about 1e-2.)"""
    plain, bwd_only = report["bf16-plain"], report["bf16-bwd"]
    assert plain["y_hat"] < 2e-2 and plain["pi"] < 2e-2
    assert plain["smooth_grad_whole"] > 2e-2
    assert _worst(bwd_only["smooth_grad_per_tensor"])[1] < 2e-2 and _worst(bwd_only["loss_grad_per_tensor"])[1] < 2e-2
