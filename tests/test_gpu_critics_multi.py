"""Several critic passes per library call (``hivt_critics_fwd`` / ``hivt_critics_bwd``) against one call per pass:
scores, input-gradient norms and every gradient, in both operand precisions; and the adversarial loss terms
computed either way."""
import pytest
import torch

import hivt_gan_amd as hv
from hivt_gan_amd import critics as critics_mod
from hivt_gan_amd import ops
from hivt_gan_amd.critics import LongScaleCritic, MidScaleCritic, ShortScaleCritic
from hivt_gan_amd.losses import AdversarialDiscriminatorLoss, AdversarialGeneratorLoss

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def make_critics(dropout):
    torch.manual_seed(3)
    cs = {"short": ShortScaleCritic(horizon=10), "mid": MidScaleCritic(horizon=20), "long": LongScaleCritic(horizon=30)}
    for c in cs.values():
        c.to(DEV).train()
        for mod in c.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = dropout
    return cs


def rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("B", [2615, 70])
def test_multi_pass_call_equals_single_pass_calls(precision, tol, B):
    prev = ops.set_matmul_precision(precision)
    try:
        gen = torch.Generator().manual_seed(B)
        real = torch.randn(B, 30, 2, generator=gen).to(DEV)
        fake = torch.randn(B, 30, 2, generator=gen).to(DEV).requires_grad_(True)
        keep = (torch.rand(B, generator=gen) > 0.2).to(DEV)
        results = {}
        for multi in (False, True):
            critics_mod.MULTI = multi
            cs = make_critics(0.0)
            d_loss, d_logs = AdversarialDiscriminatorLoss(lambda_r1=1.0).forward_masked(cs, real, fake, keep)
            params = [p for c in cs.values() for p in c.parameters()]
            d_grads = torch.autograd.grad(d_loss, params)
            g_loss, _ = AdversarialGeneratorLoss().forward_masked(cs, fake, keep)
            (g_fake,) = torch.autograd.grad(g_loss, fake)
            state = [b.clone() for c in cs.values() for b in c.buffers()]
            results[multi] = (d_loss.detach(), d_logs, d_grads, g_loss.detach(), g_fake, state)
        a, b = results[False], results[True]
        assert rel(b[0], a[0]) < tol and rel(b[3], a[3]) < tol
        for k in a[1]:
            assert rel(b[1][k], a[1][k]) < tol, k
        for i, (x, y) in enumerate(zip(a[2], b[2])):
            assert rel(y, x) < tol, f"critic parameter gradient {i}"
        assert rel(b[4], a[4]) < tol
        for x, y in zip(a[5], b[5]):  # the power iterations advanced alike
            assert torch.equal(x, y)
    finally:
        critics_mod.MULTI = True
        ops.set_matmul_precision(prev)


def test_multi_pass_dropout_draws_the_masks_of_the_single_pass_calls():
    """With dropout on, the generator-side loss draws one seed per critic in the same order either way, and both
    routes derive the layers' masks from it alike: values and the gradient into the trajectories must agree."""
    B = 700
    gen = torch.Generator().manual_seed(5)
    fake = torch.randn(B, 30, 2, generator=gen).to(DEV).requires_grad_(True)
    keep = (torch.rand(B, generator=gen) > 0.1).to(DEV)
    results = {}
    try:
        for multi in (False, True):
            critics_mod.MULTI = multi
            cs = make_critics(0.1)
            torch.manual_seed(77)
            loss, logs = AdversarialGeneratorLoss().forward_masked(cs, fake, keep)
            (g,) = torch.autograd.grad(loss, fake)
            results[multi] = (loss.detach(), logs, g)
    finally:
        critics_mod.MULTI = True
    a, b = results[False], results[True]
    assert rel(b[0], a[0]) < 2e-5
    for k in a[1]:
        assert rel(b[1][k], a[1][k]) < 2e-5, k
    assert rel(b[2], a[2]) < 2e-5
    assert float((a[2] != 0).float().mean()) > 0.5  # a gradient that is mostly zeros would prove nothing
