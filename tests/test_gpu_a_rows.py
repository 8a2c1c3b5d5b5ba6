"""The row-bookkeeping kernels (``csrc/rows.hip``) and the fused loss launches against the framework operators they
replace, on seeded inputs.  Integer / index results must be bit-exact, copies and sums of two rounded products
bit-exact as well; sums over many rows within 1e-6 of the tensor's scale.

(The file sorts first on purpose: these kernels produce the indices every later kernel of a step walks, so they
are checked on their own before anything consumes their results.)"""
import pytest
import torch
import torch.nn.functional as F

import hivt_gan_amd as hv
from hivt_gan_amd import ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def batch(scenes=6, first=300):
    return hv.make_batch(scenes, first_index=first).to(DEV)


def test_batch_prepare_equals_the_framework_operators():
    b = batch()
    prep = ops.batch_prepare(b, 20)
    ang = b["rotate_angles"]
    sin, cos = torch.sin(ang), torch.cos(ang)
    rot = torch.stack([cos, -sin, sin, cos], dim=-1).view(-1, 2, 2)
    assert torch.equal(prep["rotate_mat"], rot)
    y = b["y"]
    assert torch.equal(prep["y"], y[..., 0:1] * rot[:, None, 0, :] + y[..., 1:2] * rot[:, None, 1, :])
    x = b["x"][:, :20].transpose(0, 1)
    assert torch.equal(prep["x_steps"].view(20, -1, 2), x)
    assert torch.equal(prep["x_rot"].view(20, -1, 2), x[..., 0:1] * rot[:, 0, :] + x[..., 1:2] * rot[:, 1, :])
    code = (b["is_intersections"].long() * 3 + b["turn_directions"].long()) * 2 + b["traffic_controls"].long()
    assert torch.equal(prep["lane_code"].long(), code)
    keep = (~b["padding_mask"][:, 20:]).any(dim=1)
    assert torch.equal(prep["keep_w"], keep.float())


@pytest.mark.parametrize("capacity_share", [1.3, 1.0, 0.6])
def test_row_subset_fixed_equals_the_operator_form(capacity_share):
    b = batch(7, 340)
    pm = b["padding_mask"]
    n, T = pm.shape[0], 20
    valid = ~pm[:, :T].t().reshape(-1)
    live = int(valid.sum())
    capacity = min(max(int(live * capacity_share), 1), n * T)
    # a CSR over all T * N nodes in which padded nodes own no edges
    deg = torch.randint(0, 4, (n * T,), device=DEV, dtype=torch.int32) * valid.int()
    rowptr = torch.cat([torch.zeros(1, dtype=torch.int32, device=DEV), deg.cumsum(0).int()])
    ref = ops.row_subset(valid, capacity, rowptr)
    counts = [torch.tensor([5], dtype=torch.int32, device=DEV), torch.tensor([9], dtype=torch.int32, device=DEV)]
    got = ops.row_subset_fixed(pm, T, capacity, rowptr, counts, (5, 9))
    assert int(got.count) == live == int(ref.count)
    assert torch.equal(got.rows, ref.rows)
    assert torch.equal(got.new_id, ref.new_id)
    assert torch.equal(got.live, ref.live)
    assert torch.equal(got.rowptr, ref.rowptr)
    assert bool(got.overflow) == bool(ref.overflow) == (live > capacity)
    over = ops.row_subset_fixed(pm, T, capacity, rowptr, counts, (5, 8))
    assert bool(over.overflow)  # an edge counter above its capacity trips the flag as well


def test_row_subset_fixed_without_a_live_node():
    pm = torch.ones(9, 50, dtype=torch.bool, device=DEV)
    rowptr = torch.zeros(9 * 20 + 1, dtype=torch.int32, device=DEV)
    ref = ops.row_subset(~pm[:, :20].t().reshape(-1), 16, rowptr)
    got = ops.row_subset_fixed(pm, 20, 16, rowptr)
    assert int(got.count) == 0 and not bool(got.overflow)
    for name in ("rows", "new_id", "live", "rowptr"):
        assert torch.equal(getattr(got, name), getattr(ref, name)), name


def test_center_rows_and_token_rows():
    b = batch(5, 360)
    prep = ops.batch_prepare(b, 20)
    pm = b["padding_mask"]
    n, T = pm.shape[0], 20
    rows = ops.row_subset_fixed(pm, T, n * T, None)
    xc, code = ops.center_rows(prep["x_rot"], b["bos_mask"], rows.rows, n, T)
    ids = rows.rows.long()
    assert torch.equal(xc, prep["x_rot"][ids])
    step = ids // n
    bos = b["bos_mask"].t().reshape(-1)[ids]
    assert torch.equal(code.long(), torch.where(bos, step, torch.full_like(step, -1)))
    torch.manual_seed(3)
    x = torch.randn(n * T, 64, device=DEV, requires_grad=True)
    token = torch.randn(T, 64, device=DEV, requires_grad=True)
    out = ops.rows_put_tokens(x, code, token)
    ref = torch.where((code >= 0).unsqueeze(-1), token[code.clamp(min=0).long()], x)
    assert torch.equal(out, ref)
    g = torch.randn_like(out)
    gx, gt = torch.autograd.grad(out, (x, token), g)
    rx, rt = torch.autograd.grad(ref, (x, token), g)
    assert torch.equal(gx, rx)
    assert float((gt - rt).abs().max()) <= 1e-5 * float(rt.abs().max())


@pytest.mark.parametrize("packed", [True, False])
def test_temporal_input_equals_the_operator_form(packed):
    b = batch(4, 380)
    pm = b["padding_mask"][:, :20]
    n, T, d = pm.shape[0], 20, 64
    torch.manual_seed(5)
    pad = torch.randn(T, d, device=DEV, requires_grad=True)
    cls = torch.randn(d, device=DEV, requires_grad=True)
    pos = torch.randn(T + 1, d, device=DEV, requires_grad=True)
    if packed:
        rows = ops.row_subset_fixed(b["padding_mask"], T, int(0.8 * n * T), None)
        x = torch.randn(rows.capacity, d, device=DEV, requires_grad=True)
        ref = ops.append_token_add_steps(ops.expand_rows_with_tokens(x, rows, pm.t(), pad), cls, pos)
    else:
        rows = None
        x = torch.randn(T, n, d, device=DEV, requires_grad=True)
        ref = ops.append_token_add_steps(ops.masked_step_tokens(x, pm.t(), pad), cls, pos)
    out = ops.temporal_input(x, rows, pm, pad, cls, pos)
    assert torch.equal(out, ref)
    g = torch.randn_like(out)
    got = torch.autograd.grad(out, (x, pad, cls, pos), g)
    want = torch.autograd.grad(ref, (x, pad, cls, pos), g)
    assert torch.equal(got[0], want[0])
    for a, r in zip(got[1:], want[1:]):
        assert float((a - r).abs().max()) <= 1e-5 * float(r.abs().max())


@pytest.mark.parametrize("agent_major", [False, True])
def test_decoder_output_equals_the_operator_form(agent_major):
    torch.manual_seed(7)
    modes, n, h = 6, 37, 30
    loc = torch.randn(modes * n, 2 * h, device=DEV, requires_grad=True)
    raw = (2.0 * torch.randn(modes * n, 2 * h, device=DEV) - 1.0).requires_grad_(True)
    out = ops.decoder_output(loc, raw, modes, agent_major)

    def order(t):
        t = t.view(n, modes, h, 2).transpose(0, 1) if agent_major else t.view(modes, n, h, 2)
        return t
    ref = torch.cat((order(loc), torch.clamp(F.elu(order(raw)) + 1.0, min=0.05)), dim=-1)
    assert out.shape == (modes, n, h, 4)
    assert float((out - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    g = torch.randn_like(out)
    got = torch.autograd.grad(out, (loc, raw), g)
    want = torch.autograd.grad(ref, (loc, raw), g)
    for a, r in zip(got, want):
        assert float((a - r).abs().max()) <= 1e-6 * float(r.abs().max())


def test_best_mode_trajectories_out_of_the_loss_node():
    torch.manual_seed(9)
    b = batch(5, 420)
    n = b["x"].shape[0]
    y_hat = torch.randn(6, n, 30, 4, device=DEV, requires_grad=True)
    pi = torch.randn(n, 6, device=DEV, requires_grad=True)
    y = torch.randn(n, 30, 2, device=DEV)
    losses, best, traj = ops.supervised_loss_packed(y_hat, pi, y, b["padding_mask"], 20)
    reg, cls, best_ref = ops.supervised_loss(y_hat, pi, y, b["padding_mask"], 20)
    assert torch.equal(best, best_ref) and torch.equal(losses, torch.stack([reg, cls]))
    ref = y_hat[best, torch.arange(n, device=DEV), :, :2]
    assert torch.equal(traj, ref)
    g = torch.randn_like(traj)
    got = torch.autograd.grad((losses.sum() + (traj * g).sum()), (y_hat, pi))
    want = torch.autograd.grad(((reg + cls) + (ref * g).sum()), (y_hat, pi))
    for a, r in zip(got, want):
        assert float((a - r).abs().max()) <= 1e-6 * max(float(r.abs().max()), 1e-6)


def test_category_table_and_padded_row_sums():
    torch.manual_seed(13)
    a = torch.randn(2, 128, device=DEV, requires_grad=True)
    bb = torch.randn(3, 128, device=DEV, requires_grad=True)
    c = torch.randn(2, 128, device=DEV, requires_grad=True)
    out = ops.category_table(a, bb, c)
    ref = ((a.view(2, 1, 1, -1) + bb.view(1, 3, 1, -1)) + c.view(1, 1, 2, -1)).reshape(12, -1)
    assert torch.equal(out, ref)
    g = torch.randn_like(out)
    for x, r in zip(torch.autograd.grad(out, (a, bb, c), g), torch.autograd.grad(ref, (a, bb, c), g)):
        assert float((x - r).abs().max()) <= 1e-6 * float(r.abs().max())


def test_fused_objectives_equal_the_per_scale_launches():
    torch.manual_seed(17)
    b = 2615
    w = (torch.rand(b, device=DEV) > 0.2).float()
    cnt = w.sum().clamp(min=1.0)
    rows = [torch.randn(b, device=DEV, requires_grad=True) for _ in range(9)]
    terms = [(rows[3 * k], rows[3 * k + 1].abs(), rows[3 * k + 2]) for k in range(3)]
    total, logs = ops.hinge_r1_multi(terms, w, None, 0.7)
    ref_terms = [ops.hinge_r1_loss(s, n2, f, w, cnt, 0.7) for s, n2, f in terms]
    ref_total = (0.0 + ref_terms[0][0]) + ref_terms[1][0] + ref_terms[2][0]
    assert float((total - ref_total).abs()) <= 1e-6 * float(ref_total.abs())
    for k in range(3):
        assert float((logs[2 * k] - ref_terms[k][0]).abs()) <= 1e-6 * float(ref_terms[k][0].abs())
        assert float((logs[2 * k + 1] - ref_terms[k][1]).abs()) <= 1e-6 * float(ref_terms[k][1].abs())
    got = torch.autograd.grad(total, rows, retain_graph=True)  # (the abs nodes feed both graphs)
    want = torch.autograd.grad(ref_total, rows)
    for a, r in zip(got, want):
        assert float((a - r).abs().max()) <= 1e-6 * max(float(r.abs().max()), 1e-12)
    losses = torch.randn(2, device=DEV, requires_grad=True)
    scores = [torch.randn(b, device=DEV, requires_grad=True) for _ in range(3)]
    g_total, parts = ops.generator_total(losses, scores, w, None, 0.1)
    adv = 0.1 * ((0.0 + ops.weighted_mean(scores[0], w, cnt, -1.0)) + ops.weighted_mean(scores[1], w, cnt, -1.0)
                 + ops.weighted_mean(scores[2], w, cnt, -1.0))
    ref = (losses[0] + losses[1]) + adv
    assert float((g_total - ref).abs()) <= 1e-6 * float(ref.abs())
    assert float((parts[0] - adv).abs()) <= 1e-6 * float(adv.abs())
    for a, r in zip(torch.autograd.grad(g_total, [losses] + scores), torch.autograd.grad(ref, [losses] + scores)):
        assert float((a - r).abs().max()) <= 1e-6 * max(float(r.abs().max()), 1e-12)
