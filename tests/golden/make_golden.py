"""Generate golden vectors by running the REFERENCE's own code (run in the authoring container only).

Imports the reference's ``losses`` package and its critic classes from ``/root/reference`` (these
are torch-only; the edge-attention modules need torch_geometric and cannot be imported), feeds them
seeded inputs and stores inputs + outputs as plain arrays in ``tests/golden/reference_losses.npz``.
Nothing of the reference's source is copied; the fixture holds data only.

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_losses.npz")


def load_reference():
    sys.path.insert(0, REF)
    losses = importlib.import_module("losses")
    # the critics use relative imports; give them a package object that does not run models/__init__.py
    pkg = types.ModuleType("refcritics")
    pkg.__path__ = [os.path.join(REF, "models", "critics")]
    sys.modules["refcritics"] = pkg
    short = importlib.import_module("refcritics.short_scale_critic").ShortScaleCritic
    mid = importlib.import_module("refcritics.mid_scale_critic").MidScaleCritic
    long_ = importlib.import_module("refcritics.long_scale_critic").LongScaleCritic
    return losses, short, mid, long_


def no_dropout(module):
    for m in module.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return module


def main():
    losses, Short, Mid, Long = load_reference()
    g = torch.Generator().manual_seed(1234)
    out = {}

    # ---- LaplaceNLLLoss --------------------------------------------------------------------
    pred = torch.randn(37, 4, generator=g)
    pred[:, 2:] *= 6.0  # some log-scales beyond the [-10, 10] clamp
    pred[3, 2], pred[5, 3] = 10.0, -10.0  # exactly on the clamp boundaries
    target = torch.randn(37, 2, generator=g)
    target[7] = pred[7, :2]  # zero residual (sign(0) = 0)
    for red in ("mean", "sum", "none"):
        p = pred.clone().requires_grad_(True)
        val = losses.LaplaceNLLLoss(reduction=red)(p, target)
        val.sum().backward()
        out[f"laplace_{red}"] = val.detach().numpy()
        out[f"laplace_{red}_grad"] = p.grad.numpy()
    out["laplace_pred"], out["laplace_target"] = pred.numpy(), target.numpy()

    # ---- SoftTargetCrossEntropyLoss --------------------------------------------------------
    logits = torch.randn(11, 6, generator=g) * 3
    soft = torch.softmax(torch.randn(11, 6, generator=g), dim=-1)
    for red in ("mean", "sum", "none"):
        p = logits.clone().requires_grad_(True)
        val = losses.SoftTargetCrossEntropyLoss(reduction=red)(p, soft)
        val.sum().backward()
        out[f"ce_{red}"] = val.detach().numpy()
        out[f"ce_{red}_grad"] = p.grad.numpy()
    out["ce_logits"], out["ce_soft"] = logits.numpy(), soft.numpy()

    # ---- critics (spectral norm, training mode, dropout p=0) + hinge / R1 losses -----------
    torch.manual_seed(77)
    critics = {"short": no_dropout(Short(horizon=10, input_dim=2, hidden_dim=16)),
               "mid": no_dropout(Mid(horizon=20, input_dim=2, hidden_dim=24)),
               "long": no_dropout(Long(horizon=30, input_dim=2, hidden_dim=24))}
    prefix = {"short": "D_short", "mid": "D_mid", "long": "D_long"}
    for name, c in critics.items():
        c.train()
        for k, v in c.state_dict().items():
            out[f"critic_init.{prefix[name]}.{k}"] = v.detach().clone().numpy()
    real = torch.randn(9, 30, 2, generator=g).cumsum(dim=1)
    fake = (real + 0.5 * torch.randn(9, 30, 2, generator=g)).requires_grad_(True)
    out["real"], out["fake"] = real.numpy(), fake.detach().numpy()

    # plain forwards: two consecutive training-mode calls pin the power-iteration state evolution
    for name, c in critics.items():
        out[f"critic_fwd1.{name}"] = c(real).detach().numpy()
        out[f"critic_fwd2.{name}"] = c(real).detach().numpy()
        for k, v in c.state_dict().items():
            if k.endswith("_u") or k.endswith("_v"):
                out[f"critic_after2.{prefix[name]}.{k}"] = v.detach().clone().numpy()

    # discriminator loss (hinge + lambda_r1 * R1) and its parameter gradients
    h = {"short": 10, "mid": 20, "long": 30}
    real_trajs = {n: real[:, :h[n]].clone() for n in critics}
    fake_trajs = {n: fake[:, :h[n]] for n in critics}
    d_loss, d_logs = losses.AdversarialDiscriminatorLoss(lambda_r1=10.0)(critics, real_trajs, fake_trajs)
    params = {f"{prefix[n]}.{k}": p for n, c in critics.items() for k, p in c.named_parameters()}
    grads = torch.autograd.grad(d_loss, list(params.values()), allow_unused=True)
    out["d_loss"] = d_loss.detach().numpy()
    for k, v in d_logs.items():
        out[f"d_logs.{k}"] = v.numpy()
    for k, gr in zip(params.keys(), grads):
        out[f"d_grad.{k}"] = gr.numpy()
    for name, c in critics.items():
        for k, v in c.state_dict().items():
            if k.endswith("_u") or k.endswith("_v"):
                out[f"critic_after_d.{prefix[name]}.{k}"] = v.detach().clone().numpy()

    # generator adversarial loss and its gradient w.r.t. the fake trajectories
    g_loss, g_logs = losses.AdversarialGeneratorLoss(lambda_adv=0.1)(critics, fake_trajs)
    (g_fake,) = torch.autograd.grad(g_loss, fake)
    out["g_loss"] = g_loss.detach().numpy()
    out["g_grad_fake"] = g_fake.numpy()
    for k, v in g_logs.items():
        out[f"g_logs.{k}"] = v.numpy()

    # R1 alone on one critic
    c = critics["mid"]
    r = real[:, :20].clone().requires_grad_(True)
    r1 = losses.R1Regularization()(c(r), r)
    out["r1_mid_alone"] = r1.detach().numpy()

    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
