"""GPU parity of the reference-shaped modules and of the whole G + D + R1 step against the CPU
oracle fed with the same state_dict and the same synthetic batch (dropout 0)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import hivt_gan_amd as hv
from oracle import hivt_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rel_err(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp(min=1e-30)).item()


def assert_close(a, b, tol=1e-5, msg=""):
    a_, b_ = a.detach().cpu().double(), b.detach().cpu().double()
    assert a_.shape == b_.shape, f"{msg}: shape {tuple(a_.shape)} vs {tuple(b_.shape)}"
    if b_.numel() == 0:
        return
    err = (a_ - b_).abs().max().item()
    bound = tol * max(b_.abs().max().item(), 1e-3)
    assert err <= bound, f"{msg}: max abs err {err:.3e} > {bound:.3e} (rel-to-max {err / max(b_.abs().max().item(), 1e-30):.2e})"


def cpu_sd(module, prefix="m"):
    return {f"{prefix}.{k}": v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point)
            for k, v in module.state_dict().items()}


# A bias added to every key shifts all scores of a (target, head) row by the same amount, and the
# segment softmax is invariant to that: the exact gradient of these biases is zero and both the
# oracle and the kernels hold only rounding noise there.
# The same holds for the last bias of the mode-score head: it shifts all F logits of an agent alike
# and the loss sees them only through log_softmax over modes.
SHIFT_INVARIANT = ("lin_k.bias", "lin_k_node.bias", "lin_k_edge.bias", "decoder.pi.6.bias")


def is_shift_invariant(name):
    return name.endswith(SHIFT_INVARIANT)


def compare_param_grads(module, sd, prefix="m", tol=1e-5):
    params = dict(module.named_parameters())
    scale = max(float(sd[f"{prefix}.{k}"].grad.abs().max()) for k in params if sd[f"{prefix}.{k}"].grad is not None)
    checked, bad = 0, []
    for k, p in params.items():
        ref = sd[f"{prefix}.{k}"].grad
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, f"missing grad for {k}"
        if is_shift_invariant(k):
            assert float(p.grad.abs().max()) <= tol * scale and float(ref.abs().max()) <= tol * scale, k
            continue
        try:
            assert_close(p.grad, ref, tol, msg=k)
        except AssertionError as exc:
            bad.append(str(exc))
        checked += 1
    assert not bad, "\n".join(bad)
    assert checked > 0


def batch_tensors(b):
    return {k: b[k] for k in b.keys()}


# The model-level comparisons run the oracle in float64 on the same weights and scenes: the 1e-5 bound
# then measures the GPU path alone (a float32 CPU run carries rounding of its own of the same size).
def batch64(b):
    return {k: (b[k].double() if torch.is_tensor(b[k]) and b[k].dtype == torch.float32 else b[k]) for k in b.keys()}


def state64(state, grad_for=None):
    out = {}
    for k, v in state.items():
        t = v.detach().cpu()
        t = t.double() if t.dtype.is_floating_point else t.clone()
        if t.dtype.is_floating_point and (grad_for is None or grad_for(k)):
            t.requires_grad_(True)
        out[k] = t
    return out


@pytest.mark.parametrize("D,H,scenes", [(64, 8, 3), (128, 8, 2), (16, 2, 2)])
def test_aa_encoder_per_timestep_and_batched(D, H, scenes):
    torch.manual_seed(2022)
    b = hv.make_batch(scenes)
    data = batch_tensors(b)
    ang = data["rotate_angles"]
    rot = torch.stack([torch.cos(ang), -torch.sin(ang), torch.sin(ang), torch.cos(ang)], -1).view(-1, 2, 2)
    mod = hv.AAEncoder(historical_steps=20, node_dim=2, edge_dim=2, embed_dim=D, num_heads=H, dropout=0.0).to(DEV)
    mod.train()
    sd = cpu_sd(mod)
    # per-timestep call with the reference's arguments
    for t in (0, 7, 19):
        ei, ea = O.temporal_edges(data, t, 50.0)
        ref = O.aa_encoder(sd, "m", data["x"][:, t], t, ei, ea, data["bos_mask"][:, t], rot, 20, H)
        out = mod(x=data["x"][:, t].to(DEV), t=t, edge_index=ei.to(DEV), edge_attr=ea.to(DEV),
                  bos_mask=data["bos_mask"][:, t].to(DEV), rotate_mat=rot.to(DEV))
        assert_close(out, ref, msg=f"t={t}")
    # batched over all timesteps, forward + backward
    n = data["x"].shape[0]
    eis, eas = [], []
    for t in range(20):
        ei, ea = O.temporal_edges(data, t, 50.0)
        eis.append(ei + t * n)
        eas.append(ea)
    ei, ea = torch.cat(eis, 1), torch.cat(eas, 0)
    x_all = data["x"].transpose(0, 1).reshape(20 * n, 2)
    w = torch.randn(20 * n, D)
    ref = O.aa_encoder(sd, "m", x_all, None, ei, ea, data["bos_mask"], rot, 20, H, parallel=True)
    (ref * w).sum().backward()
    out = mod(x=x_all.to(DEV), t=None, edge_index=ei.to(DEV), edge_attr=ea.to(DEV), bos_mask=data["bos_mask"].to(DEV),
              rotate_mat=rot.to(DEV))
    (out * w.to(DEV)).sum().backward()
    assert_close(out, ref, msg="batched")
    compare_param_grads(mod, sd)
    # without rotation
    ref2 = O.aa_encoder(sd, "m", data["x"][:, 5], 5, *O.temporal_edges(data, 5, 50.0), data["bos_mask"][:, 5], None, 20, H)
    ei5, ea5 = O.temporal_edges(data, 5, 50.0)
    out2 = mod(x=data["x"][:, 5].to(DEV), t=5, edge_index=ei5.to(DEV), edge_attr=ea5.to(DEV),
               bos_mask=data["bos_mask"][:, 5].to(DEV), rotate_mat=None)
    assert_close(out2, ref2, msg="no-rotate")


@pytest.mark.parametrize("D,H,random_attrs", [(64, 8, True), (128, 8, False)])
def test_al_encoder(D, H, random_attrs):
    torch.manual_seed(7)
    b = hv.make_batch(3, random_lane_attrs=random_attrs)
    data = batch_tensors(b)
    n = data["x"].shape[0]
    ang = data["rotate_angles"]
    rot = torch.stack([torch.cos(ang), -torch.sin(ang), torch.sin(ang), torch.cos(ang)], -1).view(-1, 2, 2)
    mod = hv.ALEncoder(node_dim=2, edge_dim=2, embed_dim=D, num_heads=H, dropout=0.0).to(DEV)
    mod.train()
    sd = cpu_sd(mod)
    x_actor = torch.randn(n, D)
    ei, ea = O.distance_drop_edge(data["lane_actor_index"], data["lane_actor_vectors"], 30.0)
    xa_c = x_actor.clone().requires_grad_(True)
    ref = O.al_encoder(sd, "m", data["lane_vectors"], xa_c, ei, ea, data["is_intersections"],
                       data["turn_directions"], data["traffic_controls"], rot, H)
    w = torch.randn(n, D)
    (ref * w).sum().backward()
    xa_g = x_actor.to(DEV).requires_grad_(True)
    out = mod(x=(data["lane_vectors"].to(DEV), xa_g), edge_index=ei.to(DEV), edge_attr=ea.to(DEV),
              is_intersections=data["is_intersections"].to(DEV), turn_directions=data["turn_directions"].to(DEV),
              traffic_controls=data["traffic_controls"].to(DEV), rotate_mat=rot.to(DEV))
    (out * w.to(DEV)).sum().backward()
    assert_close(out, ref)
    assert_close(xa_g.grad, xa_c.grad, msg="d x_actor")
    compare_param_grads(mod, sd)
    # empty lane set: edge_index [2, 0]
    out0 = mod(x=(torch.zeros(0, 2, device=DEV), xa_g.detach()), edge_index=torch.zeros(2, 0, dtype=torch.long, device=DEV),
               edge_attr=torch.zeros(0, 2, device=DEV), is_intersections=torch.zeros(0, dtype=torch.uint8, device=DEV),
               turn_directions=torch.zeros(0, dtype=torch.uint8, device=DEV),
               traffic_controls=torch.zeros(0, dtype=torch.uint8, device=DEV), rotate_mat=rot.to(DEV))
    ref0 = O.al_encoder(sd, "m", torch.zeros(0, 2), x_actor, torch.zeros(2, 0, dtype=torch.long), torch.zeros(0, 2),
                        torch.zeros(0, dtype=torch.uint8), torch.zeros(0, dtype=torch.uint8),
                        torch.zeros(0, dtype=torch.uint8), rot, H)
    assert_close(out0, ref0, msg="empty lanes")


@pytest.mark.parametrize("D,H", [(64, 8), (128, 8)])
def test_global_interactor_layer_and_stack(D, H):
    torch.manual_seed(9)
    b = hv.make_batch(4)
    data = batch_tensors(b)
    n = data["x"].shape[0]
    layer = hv.GlobalInteractorLayer(embed_dim=D, num_heads=H, dropout=0.0).to(DEV)
    layer.apply(hv.init_weights)
    layer.train()
    sd = cpu_sd(layer)
    ei = O.subgraph(~data["padding_mask"][:, 19], data["edge_index"])
    x = torch.randn(n, D)
    r = torch.randn(ei.shape[1], D)
    xc, rc = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    ref = O.global_interactor_layer(sd, "m", xc, ei, rc, H)
    w = torch.randn(n, D)
    (ref * w).sum().backward()
    xg, rg = x.to(DEV).requires_grad_(True), r.to(DEV).requires_grad_(True)
    out = layer(xg, ei.to(DEV), rg)
    (out * w.to(DEV)).sum().backward()
    assert_close(out, ref)
    assert_close(xg.grad, xc.grad, msg="dx")
    assert_close(rg.grad, rc.grad, msg="d edge_attr")
    compare_param_grads(layer, sd)

    # the whole interactor on a batch (edges built on the device)
    gi = hv.GlobalInteractor(historical_steps=20, embed_dim=D, edge_dim=2, num_modes=6, num_heads=H, num_layers=3,
                             dropout=0.0, rotate=True).to(DEV)
    gi.train()
    sdg = cpu_sd(gi)
    ang = data["rotate_angles"]
    rot = torch.stack([torch.cos(ang), -torch.sin(ang), torch.sin(ang), torch.cos(ang)], -1).view(-1, 2, 2)
    dc = dict(data)
    dc["rotate_mat"] = rot
    le = torch.randn(n, D)
    lec = le.clone().requires_grad_(True)
    refg = O.global_interactor(sdg, "m", dc, lec, 20, 6, 3, H)
    wg = torch.randn(6, n, D)
    (refg * wg).sum().backward()
    g = b.to(DEV)
    g["rotate_mat"] = rot.to(DEV)
    leg = le.to(DEV).requires_grad_(True)
    outg = gi(g, leg)
    (outg * wg.to(DEV)).sum().backward()
    assert_close(outg, refg)
    assert_close(leg.grad, lec.grad, msg="d local_embed")
    compare_param_grads(gi, sdg)


CFG64 = dict(historical_steps=20, future_steps=30, num_modes=6, rotate=True, node_dim=2, edge_dim=2, embed_dim=64,
             num_heads=8, dropout=0.0, num_temporal_layers=4, num_global_layers=3, local_radius=50.0, parallel=False)


def test_hivt64_generator_forward_backward_vs_oracle():
    """HiVT-64 generator forward + supervised loss + backward on a synthetic batch: the GPU path must
    match the reference-faithful per-timestep CPU path in outputs, losses and every parameter gradient."""
    torch.manual_seed(2022)
    model = hv.HiVT(**CFG64, lr=5e-4, weight_decay=1e-4, T_max=64).to(DEV)
    model.train()
    b = hv.make_batch(8)
    sd = state64(model.state_dict())
    y_ref, pi_ref, y_rot = O.hivt_forward(sd, batch64(b), dict(CFG64), training=False)
    reg_ref, cls_ref, best_ref = O.supervised_losses(y_ref, pi_ref, y_rot, b["padding_mask"], 20)
    (reg_ref + cls_ref).backward()
    g = b.to(DEV)
    y_hat, pi = model(g)
    assert_close(g["y"], y_rot, msg="rotated targets")
    reg, cls, best = model._supervised_losses(g, y_hat, pi)
    (reg + cls).backward()
    assert_close(y_hat, y_ref, msg="y_hat")
    assert_close(pi, pi_ref, msg="pi")
    assert_close(reg, reg_ref, msg="reg")
    assert_close(cls, cls_ref, msg="cls")
    valid = (~b["padding_mask"][:, 20:]).any(dim=1)
    assert torch.equal(best.cpu()[valid], best_ref[valid])
    worst = 0.0
    bad = []
    for k, p in model.named_parameters():
        ref = sd[k].grad
        if ref is None or is_shift_invariant(k):
            continue
        e = rel_err(p.grad, ref) if float(ref.abs().max()) > 0 else float(p.grad.abs().max())
        worst = max(worst, e)
        if e >= 1e-5:
            bad.append(f"{k}: relative error {e:.2e}")
    assert not bad, "\n".join(bad)
    print(f"worst parameter-gradient relative error {worst:.2e}")


def test_dense_scene_generator_vs_oracle():
    """Dense-scene stress shape (200 agents, 1000 lane segments, 150 m radius, every step valid): 796 000
    agent-agent edges with 199 per target row, 200 000 lane-actor edges with 1000 per row.  HiVT-64
    generator forward, supervised losses and every parameter gradient against the per-timestep CPU path."""
    cfg = dict(CFG64, local_radius=150.0)
    torch.manual_seed(2022)
    model = hv.HiVT(**cfg, lr=5e-4, weight_decay=1e-4, T_max=64).to(DEV)
    model.train()
    b = hv.make_batch(1, num_agents=200, num_lanes=1000, all_valid=True, local_radius=150.0)
    assert b["edge_index"].shape[1] == 200 * 199 and b["lane_actor_index"].shape[1] > 100_000
    sd = state64(model.state_dict())
    y_ref, pi_ref, y_rot = O.hivt_forward(sd, batch64(b), dict(cfg), training=False)
    reg_ref, cls_ref, best_ref = O.supervised_losses(y_ref, pi_ref, y_rot, b["padding_mask"], 20)
    (reg_ref + cls_ref).backward()
    g = b.to(DEV)
    y_hat, pi = model(g)
    plan = g["edge_plan"]
    assert plan.agent_agent.num_edges > 700_000 and plan.lane_actor.num_edges > 100_000
    reg, cls, best = model._supervised_losses(g, y_hat, pi)
    (reg + cls).backward()
    assert_close(y_hat, y_ref, msg="y_hat")
    assert_close(pi, pi_ref, msg="pi")
    assert_close(reg, reg_ref, msg="reg")
    assert_close(cls, cls_ref, msg="cls")
    assert torch.equal(best.cpu(), best_ref)
    bad = []
    for k, p in model.named_parameters():
        ref = sd[k].grad
        if ref is None or is_shift_invariant(k):
            continue
        e = rel_err(p.grad, ref) if float(ref.abs().max()) > 0 else float(p.grad.abs().max())
        if e >= 1e-5:
            bad.append(f"{k}: relative error {e:.2e}")
    assert not bad, "\n".join(bad)


def test_gan_step_gradients_vs_oracle():
    """One G + D + R1 step (critic dropout 0): critic and generator gradients and all losses."""
    torch.manual_seed(2022)
    model = hv.HiVT(**CFG64, lr=5e-4, weight_decay=1e-4, T_max=64, use_gan=True, lambda_adv=0.1, lambda_r1=1.0).to(DEV)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    b = hv.make_batch(6, first_index=100)
    full = model.state_dict()
    gen_sd = state64({k: v for k, v in full.items() if not k.startswith("D_")})
    cr_sd = state64({k: v for k, v in full.items() if k.startswith("D_")},
                    grad_for=lambda k: k.endswith("weight_orig") or k.endswith("bias"))
    cfg = dict(CFG64, lambda_adv=0.1, lambda_r1=1.0, critic_dropout=0.0)

    # lr = 0 so the critic update between the two backwards leaves the weights in place
    class Frozen:
        def __init__(self):
            self.grads = {}

        def zero_grad(self, set_to_none=False):
            pass

        def step(self):
            pass

    opt_g, opt_d = Frozen(), Frozen()
    ref = O.gan_step_grads(gen_sd, cr_sd, batch64(b), cfg, training=True)
    g = b.to(DEV)
    grabbed = {}

    def grab(group, bad=None):
        params = model.critic_parameters() if group == "critic" else model.generator_parameters()
        names = {id(p): k for k, p in model.named_parameters()}
        grabbed[group] = {names[id(p)]: p.grad.detach().clone() for p in params if p.grad is not None}
        if group == "critic":
            for p in params:
                p.grad = None

    logs = model.training_step(g, opt_g, opt_d, sync_grads=grab)
    assert_close(logs["reg_loss"], ref["reg_loss"], msg="reg")
    assert_close(logs["cls_loss"], ref["cls_loss"], msg="cls")
    assert_close(logs["d_loss"], ref["d_loss"], msg="d_loss")
    assert_close(logs["g_adv"], ref["g_adv"], msg="g_adv")
    for k, gref in ref["d_grads"].items():
        assert rel_err(grabbed["critic"][k], gref) < 1e-5, (k, rel_err(grabbed["critic"][k], gref))
    worst = 0.0
    bad = []
    for k, gref in ref["g_grads"].items():
        if gref is None or is_shift_invariant(k):
            continue
        e = rel_err(grabbed["generator"][k], gref)
        worst = max(worst, e)
        if e >= 1e-5:
            bad.append(f"{k}: {e:.2e}")
    assert not bad, "\n".join(bad)
    print(f"worst generator-gradient relative error in the GAN step {worst:.2e}")


def test_r1_double_backward_coexists_with_custom_functions():
    """create_graph=True through the critics while the generator graph (custom Functions) is alive."""
    torch.manual_seed(1)
    model = hv.HiVT(**CFG64, lr=5e-4, weight_decay=1e-4, T_max=64, use_gan=True).to(DEV)
    model.train()
    trainer = hv.GanTrainer(model, distributed=False)
    b = hv.make_batch(4).to(DEV)
    y0 = b["y"]
    before = [p.detach().clone() for p in model.critic_parameters()[:2]] + [model.generator_parameters()[0].detach().clone()]
    for _ in range(2):
        b["y"] = y0
        logs = trainer.step(b)
    assert all(torch.isfinite(v).all() for v in logs.values())
    after = [p.detach() for p in model.critic_parameters()[:2]] + [model.generator_parameters()[0].detach()]
    assert all(not torch.equal(a, c) for a, c in zip(before, after))


# ---------------------------------------------------------------------------------------------
# Critics: fused spectral normalisation
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("training", [True, False])
def test_critic_matches_framework_hook_layers(training):
    """The critic (one spectral-norm launch + one MLP call per forward) against
    torch.nn.utils.spectral_norm layers holding the same state: scores, the advanced u / v buffers, the
    R1 quantity (double backward on the framework side) and every gradient."""
    import copy
    from torch.nn.utils import spectral_norm

    torch.manual_seed(5)
    critic = hv.LongScaleCritic(horizon=30, input_dim=2, hidden_dim=256).to(DEV)
    for mod in critic.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    ref_layers = []
    for m in critic._spectral_layers():
        # the framework reference runs in float64, so the bounds below measure the library alone
        lin = spectral_norm(torch.nn.Linear(m.in_features, m.out_features)).to(DEV).double()
        lin.load_state_dict({k: v.double() for k, v in copy.deepcopy(m.state_dict()).items()})
        ref_layers.append(lin)
    critic.train(training)
    for lin in ref_layers:
        lin.train(training)

    def ref_forward(x):
        h = x.reshape(x.shape[0], -1)
        for lin in ref_layers[:-1]:
            h = F.leaky_relu(lin(h), 0.2)
        return ref_layers[-1](h).squeeze(-1)

    x = torch.randn(300, 30, 2, device=DEV)
    wts = torch.rand(300, device=DEV)
    for call in range(2):  # two calls: the second starts from the advanced vectors
        sa, na = critic.scores_and_input_grad_norm(x)
        xb = x.double().requires_grad_(True)
        sb = ref_forward(xb)
        (gb,) = torch.autograd.grad(sb.sum(), xb, create_graph=True)
        nb = gb.pow(2).sum(dim=(1, 2))
        assert_close(sa, sb, 1e-5, msg=f"scores call {call}")
        assert_close(na, nb, 1e-5, msg=f"grad norm call {call}")
        la = (F.relu(1.0 - sa) * wts).sum() + (na * wts).sum()
        lb = (F.relu(1.0 - sb) * wts.double()).sum() + (nb * wts.double()).sum()
        for p in critic.parameters():
            p.grad = None
        for lin in ref_layers:
            for p in lin.parameters():
                p.grad = None
        la.backward()
        lb.backward()
        for m, lin in zip(critic._spectral_layers(), ref_layers):
            assert_close(m.weight_u, lin.weight_u, 1e-5, msg="u")
            assert_close(m.weight_v, lin.weight_v, 1e-5, msg="v")
            assert_close(m.weight_orig.grad, lin.weight_orig.grad, 1e-5, msg="d weight_orig")
            assert_close(m.bias.grad, lin.bias.grad, 1e-5, msg="d bias")
    # gradient with respect to the trajectories (the generator's adversarial term), evaluation mode
    critic.eval()
    for lin in ref_layers:
        lin.eval()
    xa, xb = x.clone().requires_grad_(True), x.double().requires_grad_(True)
    (ga,) = torch.autograd.grad((critic(xa) * wts).sum(), xa)
    (gb,) = torch.autograd.grad((ref_forward(xb) * wts.double()).sum(), xb)
    assert_close(ga, gb, 1e-5, msg="d input")


@pytest.mark.parametrize("B,widths,p", [(777, [60, 256, 256, 256, 256], 0.3), (1, [20, 128, 128], 0.0),
                                        (2623, [40, 256, 256, 256], 0.1)])
def test_critic_entry_points_with_dropout(B, widths, p):
    """hivt_critic_fwd / hivt_critic_bwd through the C ABI, dropout on: the masks are read back from the
    saved activations and replayed through framework ops, whose double backward is the reference for
    the ||d score / d x||^2 path."""
    import ctypes
    from hivt_gan_amd import _lib
    gen = torch.Generator().manual_seed(B + len(widths))
    L = len(widths) - 1
    x = torch.randn(B, widths[0], generator=gen).to(DEV)
    ws = [(torch.randn(widths[l + 1], widths[l], generator=gen) / widths[l] ** 0.5).to(DEV) for l in range(L)]
    ws.append((torch.randn(1, widths[L], generator=gen) / widths[L] ** 0.5).to(DEV))
    bs = [(torch.randn(widths[l + 1], generator=gen) * 0.1).to(DEV) for l in range(L)] + [torch.randn(1, generator=gen).to(DEV)]
    slope = 0.2
    c_w = (ctypes.c_int32 * (L + 1))(*widths)
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    n_saved = _lib.query("hivt_critic_saved_floats", B, L, tuple(widths), 1)
    saved = torch.zeros(n_saved, device=DEV)
    scores, gn2 = torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.call("hivt_critic_fwd", x.data_ptr(), widths[0], B, L, c_w, arr(ws), arr(bs), slope, p, 1234, None, 1,
              saved.data_ptr(), scores.data_ptr(), gn2.data_ptr(), stream)
    # hidden activations are the first blocks of `saved` (each padded to 64 floats)
    hs, off = [], 0
    for l in range(1, L + 1):
        hs.append(saved[off:off + B * widths[l]].view(B, widths[l]).clone())
        off += (B * widths[l] + 63) // 64 * 64
    # framework replay with the same masks, in float64 (the bounds then measure the library alone)
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    br = [b.double().requires_grad_(True) for b in bs]
    h = xr
    for l in range(L):
        # the activation / dropout factor of every element is taken from the library's own output (its
        # sign), so that an element whose pre-activation is zero up to rounding sits on the same linear
        # piece on both sides
        factor = (torch.where(hs[l] > 0, 1.0, slope) * (hs[l] != 0).float()).double() / (1.0 - p)
        if p > 0.0:
            kept = float((hs[l] != 0).float().mean())
            assert abs(kept - (1.0 - p)) < 0.02, (l, kept)
        h = F.linear(h, wr[l], br[l]) * factor
        assert_close(hs[l], h, 1e-5, msg=f"h{l + 1}")
    sr = F.linear(h, wr[L], br[L]).squeeze(-1)
    (gx,) = torch.autograd.grad(sr.sum(), xr, create_graph=True)
    nr = gx.pow(2).sum(dim=1)
    assert_close(scores, sr, 1e-5, msg="scores")
    assert_close(gn2, nr, 1e-5, msg="grad norm")
    g_s = torch.randn(B, generator=gen).to(DEV)
    g_n = torch.randn(B, generator=gen).to(DEV)
    dws = [torch.empty_like(w) for w in ws]
    dbs = [torch.empty_like(b) for b in bs]
    dx = torch.empty(B, widths[0], device=DEV)
    wsb = torch.empty(_lib.query("hivt_critic_workspace_bytes", B, L, tuple(widths)), dtype=torch.uint8, device=DEV)
    for use_s, use_n in ((True, True), (True, False), (False, True)):
        _lib.call("hivt_critic_bwd", g_s.data_ptr() if use_s else None, g_n.data_ptr() if use_n else None, x.data_ptr(),
                  widths[0], B, L, c_w, arr(ws), slope, p, saved.data_ptr(), arr(dws), arr(dbs), dx.data_ptr(),
                  wsb.data_ptr(), wsb.numel(), stream)
        loss = (sr * g_s.double()).sum() * (1.0 if use_s else 0.0) + (nr * g_n.double()).sum() * (1.0 if use_n else 0.0)
        ref = torch.autograd.grad(loss, wr + br, retain_graph=True, allow_unused=True)
        for l in range(L + 1):
            assert_close(dws[l], ref[l], 1e-5, msg=f"dW{l} {use_s}{use_n}")
            rb = ref[L + 1 + l] if ref[L + 1 + l] is not None else torch.zeros_like(bs[l])
            assert_close(dbs[l], rb, 1e-5, msg=f"db{l} {use_s}{use_n}")
        ref_dx = torch.autograd.grad((sr * g_s.double()).sum(), xr, retain_graph=True)[0] if use_s else torch.zeros_like(x)
        assert_close(dx, ref_dx, 1e-5, msg=f"dx {use_s}{use_n}")


def test_temporal_last_position_equals_full_layer_row():
    """The last layer evaluated for the last position only gives row [-1] of the full evaluation, and the
    same gradients for every parameter and for the input sequence."""
    torch.manual_seed(3)
    layer = hv.TemporalEncoderLayer(embed_dim=128, num_heads=8, dropout=0.0).to(DEV)
    src = torch.randn(21, 500, 128, device=DEV, requires_grad=True)
    go = torch.randn(500, 128, device=DEV)
    full = layer(src)[-1]
    g_full = torch.autograd.grad(full, [src] + list(layer.parameters()), go, allow_unused=True)
    last = layer.forward_last(src)
    g_last = torch.autograd.grad(last, [src] + list(layer.parameters()), go, allow_unused=True)
    assert_close(last, full, 1e-5, msg="last row")
    names = ["src"] + [k for k, _ in layer.named_parameters()]
    for k, a, b in zip(names, g_last, g_full):
        assert (a is None) == (b is None), k
        if a is not None:
            assert_close(a, b, 3e-5, msg=k)


def test_validation_step_metrics_vs_oracle_forward():
    """validation_step on the GPU: regression loss, the target agents' best-by-endpoint trajectories and
    minADE / minFDE / minMR against the same quantities computed from the CPU path's predictions
    (models/hivt.py:347-378)."""
    torch.manual_seed(2022)
    model = hv.HiVT(**CFG64, lr=5e-4, weight_decay=1e-4, T_max=64).to(DEV)
    model.eval()
    b = hv.make_batch(6)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        y_ref, pi_ref, y_rot = O.hivt_forward(sd, batch_tensors(b), dict(CFG64), training=False)
        reg_ref, _cls, _best = O.supervised_losses(y_ref, pi_ref, y_rot, b["padding_mask"], 20)
    agent = b["agent_index"]
    ya = y_rot[agent]
    pa = y_ref[:, agent, :, :2]
    fde = (pa[:, :, -1] - ya[:, -1]).norm(dim=-1)
    best = pa[fde.argmin(dim=0), torch.arange(agent.shape[0])]
    want_ade = float((best - ya).norm(dim=-1).mean(dim=-1).mean())
    want_fde = float(fde.min(dim=0).values.mean())
    want_mr = float((fde.min(dim=0).values > 2.0).float().mean())
    vm = hv.ValidationMetrics()
    out = model.validation_step(b.to(DEV), vm)
    got = vm.compute()
    assert torch.equal(out["best_mode_agent"].cpu(), fde.argmin(dim=0))
    assert abs(float(got["val_minADE"]) - want_ade) <= 2e-5 * max(1.0, want_ade)
    assert abs(float(got["val_minFDE"]) - want_fde) <= 2e-5 * max(1.0, want_fde)
    assert abs(float(got["val_minMR"]) - want_mr) <= 1e-6
    assert abs(float(got["val_reg_loss"]) - float(reg_ref)) <= 2e-5 * max(1.0, abs(float(reg_ref)))
    assert abs(float(out["val_minADE"]) - want_ade) <= 2e-5 * max(1.0, want_ade)
    assert float(got["val_jerk"]) >= 0.0 and 0.0 <= float(got["val_speed_violation"]) <= 1.0
    assert float(got["val_endpoint_diversity"]) > 0.0
    # a second batch accumulates; a batch without any future step leaves the running loss alone
    b2 = hv.make_batch(4, first_index=50)
    model.validation_step(b2.to(DEV), vm)
    b3 = hv.make_batch(3, first_index=80)
    b3["padding_mask"][:, 20:] = True
    before = float(vm.val_reg_loss.count)
    model.validation_step(b3.to(DEV), vm)
    assert float(vm.val_reg_loss.count) == before == 2.0 and float(vm.minADE.count) == 13.0
