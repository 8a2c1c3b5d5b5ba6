"""The oracle's loss / critic restatement against vectors produced by the reference's own code
(tests/golden/make_golden.py -> tests/golden/reference_losses.npz)."""
import numpy as np
import pytest
import torch

from oracle import hivt_oracle as O

RTOL, ATOL = 1e-5, 1e-6  # fp32; same arithmetic, possibly different op fusion


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("red", ["mean", "sum", "none"])
def test_laplace_nll(golden, red):
    pred = T(golden["laplace_pred"]).clone().requires_grad_(True)
    val = O.laplace_nll(pred, T(golden["laplace_target"]), reduction=red)
    val.sum().backward()
    np.testing.assert_allclose(val.detach().numpy(), golden[f"laplace_{red}"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(pred.grad.numpy(), golden[f"laplace_{red}_grad"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("red", ["mean", "sum", "none"])
def test_soft_target_ce(golden, red):
    logits = T(golden["ce_logits"]).clone().requires_grad_(True)
    val = O.soft_target_cross_entropy(logits, T(golden["ce_soft"]), reduction=red)
    val.sum().backward()
    np.testing.assert_allclose(val.detach().numpy(), golden[f"ce_{red}"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(logits.grad.numpy(), golden[f"ce_{red}_grad"], rtol=RTOL, atol=ATOL)


def test_invalid_reduction_raises():
    with pytest.raises(ValueError):
        O.laplace_nll(torch.zeros(2, 4), torch.zeros(2, 2), reduction="bogus")
    with pytest.raises(ValueError):
        O.soft_target_cross_entropy(torch.zeros(2, 3), torch.zeros(2, 3), reduction="bogus")


def critic_state(golden):
    sd = {}
    for k, v in golden.items():
        if k.startswith("critic_init."):
            t = T(v).clone()
            name = k[len("critic_init."):]
            if name.endswith("weight_orig") or name.endswith("bias"):
                t.requires_grad_(True)
            sd[name] = t
    return sd


HORIZON = {"short": 10, "mid": 20, "long": 30}


def test_critic_forward_and_power_iteration(golden):
    sd = critic_state(golden)
    real = T(golden["real"])
    for name in ("short", "mid", "long"):
        p = O.CRITIC_PREFIX[name]
        o1 = O.critic_forward(sd, p, real, HORIZON[name], O.CRITIC_LAYERS[name], 0.0, True)
        o2 = O.critic_forward(sd, p, real, HORIZON[name], O.CRITIC_LAYERS[name], 0.0, True)
        np.testing.assert_allclose(o1.detach().numpy(), golden[f"critic_fwd1.{name}"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(o2.detach().numpy(), golden[f"critic_fwd2.{name}"], rtol=RTOL, atol=ATOL)
    for k, v in golden.items():
        if k.startswith("critic_after2."):
            np.testing.assert_allclose(sd[k[len("critic_after2."):]].numpy(), v, rtol=RTOL, atol=ATOL)


def test_discriminator_and_generator_losses(golden):
    sd = critic_state(golden)
    real, fake = T(golden["real"]), T(golden["fake"]).clone().requires_grad_(True)
    # replay the two plain forwards the fixture ran first, so the power-iteration state matches
    for name in ("short", "mid", "long"):
        for _ in range(2):
            O.critic_forward(sd, O.CRITIC_PREFIX[name], real, HORIZON[name], O.CRITIC_LAYERS[name], 0.0, True)
    real_t = {n: real[:, :HORIZON[n]].clone() for n in HORIZON}
    fake_t = {n: fake[:, :HORIZON[n]] for n in HORIZON}
    d_loss, d_logs = O.discriminator_loss(sd, real_t, fake_t, HORIZON, 10.0, 0.0, True)
    np.testing.assert_allclose(d_loss.detach().numpy(), golden["d_loss"], rtol=RTOL, atol=ATOL)
    for k, v in d_logs.items():
        np.testing.assert_allclose(v.numpy(), golden[f"d_logs.{k}"], rtol=RTOL, atol=ATOL)
    keys = [k[len("d_grad."):] for k in golden if k.startswith("d_grad.")]
    grads = torch.autograd.grad(d_loss, [sd[k] for k in keys])
    for k, g in zip(keys, grads):
        np.testing.assert_allclose(g.numpy(), golden[f"d_grad.{k}"], rtol=2e-4, atol=2e-6, err_msg=k)
    for k, v in golden.items():
        if k.startswith("critic_after_d."):
            np.testing.assert_allclose(sd[k[len("critic_after_d."):]].numpy(), v, rtol=RTOL, atol=ATOL)
    g_loss, g_logs = O.generator_adv_loss(sd, fake_t, HORIZON, 0.1, 0.0, True)
    (g_fake,) = torch.autograd.grad(g_loss, fake)
    np.testing.assert_allclose(g_loss.detach().numpy(), golden["g_loss"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(g_fake.numpy(), golden["g_grad_fake"], rtol=RTOL, atol=1e-7)
    for k, v in g_logs.items():
        np.testing.assert_allclose(v.numpy(), golden[f"g_logs.{k}"], rtol=RTOL, atol=ATOL)
    r = real[:, :20].clone().requires_grad_(True)
    r1 = O.r1_penalty(O.critic_forward(sd, "D_mid", r, 20, 3, 0.0, True), r)
    np.testing.assert_allclose(r1.detach().numpy(), golden["r1_mid_alone"], rtol=RTOL, atol=ATOL)
