"""GPU parity of each C-ABI kernel against the CPU oracle on identical seeded inputs.

Bars: bit-exact for edge sets, CSR order and indices; fp32 results within 1e-5 relative
(rtol 1e-5 plus a small absolute floor for values near zero)."""
import os

import numpy as np
import pytest
import torch

import _parity

import hivt_gan_amd as hv
from hivt_gan_amd import ops
from oracle import hivt_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
RTOL = 1e-5


def close(a, b, rtol=RTOL, atol=2e-6, msg=""):
    a = a.detach().cpu().double().numpy()
    b = b.detach().cpu().double().numpy()
    scale = max(np.abs(b).max(), 1e-30) if b.size else 1.0
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * max(scale, 1.0), err_msg=msg)


def close64(gpu, ref32, ref64, rtol=RTOL, msg=""):
    """fp32 GPU result against the fp64 oracle: within 1e-5 of the tensor's scale, or -- where the
    fp32 CPU reference itself is further than that from fp64 (ill-conditioned softmax gradients) --
    within twice the fp32 reference's own error."""
    g = gpu.detach().cpu().double()
    r64 = ref64.detach().double()
    scale = max(float(r64.abs().max()), 1e-30) if r64.numel() else 1.0
    err_gpu = float((g - r64).abs().max()) if r64.numel() else 0.0
    err_ref = float((ref32.detach().double() - r64).abs().max()) if r64.numel() else 0.0
    bound = max(rtol * scale, 2.0 * err_ref)
    _parity.record(os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], msg or "tensor", gpu, r64, bound,
                   "rtol" if err_gpu <= rtol * scale else "2x fp32 reference", max_abs_error=err_gpu, scale=scale,
                   fp32_reference_error=err_ref)
    assert err_gpu <= bound, f"{msg}: gpu err {err_gpu:.3e} > bound {bound:.3e} (scale {scale:.3e}, fp32 ref err {err_ref:.3e})"


# ---------------------------------------------------------------------------------------------
# edge construction
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scenes,radius", [(3, 50.0), (8, 20.0), (2, None)])
def test_temporal_edges_bit_exact(scenes, radius):
    b = hv.make_batch(scenes)
    g = b.to(DEV)
    T, n = 20, int(b.num_nodes)
    src, dst, attr, count = ops.edges_temporal(g["edge_index"], g["padding_mask"], g["positions"], 0, T, radius)
    e = int(count.item())
    data = {k: b[k] for k in b.keys()}
    eis, eas = [], []
    for t in range(T):
        ei, ea = O.temporal_edges(data, t, radius)
        eis.append(ei + t * n)
        eas.append(ea)
    ei, ea = torch.cat(eis, dim=1), torch.cat(eas, dim=0)
    assert e == ei.shape[1]
    assert torch.equal(src[:e].cpu().long(), ei[0])
    assert torch.equal(dst[:e].cpu().long(), ei[1])
    assert torch.equal(attr[:e].cpu(), ea)  # same subtraction, bit-exact


def test_global_edges_and_radius_filter_bit_exact():
    b = hv.make_batch(5)
    g = b.to(DEV)
    src, dst, attr, count = ops.edges_temporal(g["edge_index"], g["padding_mask"], g["positions"], 19, 1, None)
    e = int(count.item())
    ei = O.subgraph(~b["padding_mask"][:, 19], b["edge_index"])
    assert e == ei.shape[1] and torch.equal(torch.stack([src[:e], dst[:e]]).cpu().long(), ei)
    for radius in (50.0, 10.0, 1e-3, None):
        s2, d2, a2, c2 = ops.edges_radius(g["lane_actor_index"], g["lane_actor_vectors"], radius)
        e2 = int(c2.item())
        ei2, ea2 = O.distance_drop_edge(b["lane_actor_index"], b["lane_actor_vectors"], radius)
        assert e2 == ei2.shape[1]
        assert torch.equal(torch.stack([s2[:e2], d2[:e2]]).cpu().long(), ei2)
        assert torch.equal(a2[:e2].cpu(), ea2)


def test_radius_boundary_matches_host_norm():
    """Vectors whose length is within a few ulp of the radius: the kept set must equal the host's
    ``torch.norm(...) < r`` decision for every one of them."""
    gen = torch.Generator().manual_seed(5)
    n = 200_000
    ang = torch.rand(n, generator=gen, dtype=torch.float64) * 2 * np.pi
    r = 50.0 * (1.0 + (torch.rand(n, generator=gen, dtype=torch.float64) - 0.5) * 4e-7)
    attr = torch.stack([r * torch.cos(ang), r * torch.sin(ang)], dim=-1).float()
    ei = torch.stack([torch.arange(n), torch.arange(n)]).long()
    keep_host = torch.norm(attr, p=2, dim=-1) < 50.0
    assert 0.2 < keep_host.float().mean() < 0.8
    s, d, a, c = ops.edges_radius(ei.to(DEV), attr.to(DEV), 50.0)
    e = int(c.item())
    assert e == int(keep_host.sum())
    assert torch.equal(s[:e].cpu().long(), ei[0][keep_host])


def test_empty_edge_lists():
    ei = torch.zeros(2, 0, dtype=torch.long, device=DEV)
    s, d, a, c = ops.edges_radius(ei, torch.zeros(0, 2, device=DEV), 50.0)
    assert int(c.item()) == 0
    rowptr, perm = ops.csr_build(torch.zeros(0, dtype=torch.int32, device=DEV), 7)
    assert rowptr.cpu().tolist() == [0] * 8 and perm.numel() == 0


@pytest.mark.parametrize("e,rows", [(1, 1), (1000, 37), (50_000, 640), (200_000, 50), (30_000, 20_000), (40_000, 100),
                                    (60_000, 120), (141_000, 2600)])
def test_csr_is_stable_sort(e, rows):
    gen = torch.Generator().manual_seed(e + rows)
    key = torch.randint(0, rows, (e,), generator=gen, dtype=torch.int32)
    rowptr, perm = ops.csr_build(key.to(DEV), rows)
    order = torch.sort(key.long(), stable=True).indices
    assert torch.equal(perm.cpu().long(), order)
    counts = torch.bincount(key.long(), minlength=rows)
    assert torch.equal(rowptr.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]))


def test_csr_of_live_prefix_ignores_the_tail():
    # fixed-capacity lists: only the first `count` keys are grouped; the tail may hold anything
    gen = torch.Generator().manual_seed(11)
    e, rows, live = 20_000, 300, 13_337
    key = torch.randint(0, rows, (e,), generator=gen, dtype=torch.int32)
    key[live:] = 7  # in range on purpose: a kernel that looked at the tail would show up in the counts
    count = torch.tensor([live], dtype=torch.int32, device=DEV)
    rowptr, perm = ops.csr_build(key.to(DEV), rows, count)
    order = torch.sort(key[:live].long(), stable=True).indices
    assert torch.equal(perm[:live].cpu().long(), order)
    counts = torch.bincount(key[:live].long(), minlength=rows)
    assert torch.equal(rowptr.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]))
    # a count beyond the capacity is clamped to it
    rowptr2, _ = ops.csr_build(key[:live].to(DEV), rows, torch.tensor([10 * e], dtype=torch.int32, device=DEV))
    assert torch.equal(rowptr2, rowptr)


def test_csr_repeatable():
    key = torch.randint(0, 300, (100_000,), dtype=torch.int32, device=DEV)
    p0 = ops.csr_build(key, 300)[1]
    for _ in range(3):
        assert torch.equal(ops.csr_build(key, 300)[1], p0)


# ---------------------------------------------------------------------------------------------
# gather + rotate
# ---------------------------------------------------------------------------------------------
def test_gather_rotate_and_relpose():
    gen = torch.Generator().manual_seed(3)
    ns, nd, e = 50, 40, 3000
    x = torch.randn(ns, 2, generator=gen)
    attr = torch.randn(e, 2, generator=gen) * 30
    ang = torch.rand(nd, generator=gen) * 6.28
    rot = torch.stack([torch.cos(ang), -torch.sin(ang), torch.sin(ang), torch.cos(ang)], -1).view(nd, 2, 2)
    src = torch.randint(0, ns, (e,), generator=gen, dtype=torch.int32)
    dst = torch.randint(0, nd, (e,), generator=gen, dtype=torch.int32)
    rowptr, perm = ops.csr_build(dst.to(DEV), nd)
    xg = x.to(DEV).requires_grad_(True)
    ag = attr.to(DEV).requires_grad_(True)
    out, s_csr, d_csr = ops.edge_gather_rotate(xg, ag, rot.to(DEV), src.to(DEV), dst.to(DEV), perm, nd)
    p = perm.cpu().long()
    xc, ac = x.clone().requires_grad_(True), attr.clone().requires_grad_(True)
    R = rot[dst.long()[p]]
    ref = torch.cat([torch.bmm(xc[src.long()[p]].unsqueeze(-2), R).squeeze(-2),
                     torch.bmm(ac[p].unsqueeze(-2), R).squeeze(-2)], dim=-1)
    close(out, ref)
    assert torch.equal(s_csr.cpu(), src[p]) and torch.equal(d_csr.cpu(), dst[p])
    w = torch.randn(e, 4, generator=gen)
    (out * w.to(DEV)).sum().backward()
    (ref * w).sum().backward()
    close(ag.grad, ac.grad)
    close(xg.grad, xc.grad)  # rows of one source summed in a fixed order (no float atomics)
    first = xg.grad.clone()
    xg.grad = None
    out_again, _, _ = ops.edge_gather_rotate(xg, ag, rot.to(DEV), src.to(DEV), dst.to(DEV), perm, nd)
    (out_again * w.to(DEV)).sum().backward()
    assert torch.equal(xg.grad, first), "the source-row gradient must be bit-reproducible"
    # no rotation
    out2, _, _ = ops.edge_gather_rotate(x.to(DEV), attr.to(DEV), None, src.to(DEV), dst.to(DEV), perm, 1)
    assert torch.equal(out2.cpu(), torch.cat([x[src.long()[p]], attr[p]], dim=-1))
    # relative pose of the global interactor
    ang_n = torch.rand(nd, generator=gen) * 6.28 - 3.14
    srcn = torch.randint(0, nd, (e,), generator=gen, dtype=torch.int32)
    o3, _, _ = ops.edge_relpose(attr.to(DEV), rot.to(DEV), ang_n.to(DEV), srcn.to(DEV), dst.to(DEV), perm)
    dth = ang_n[srcn.long()[p]] - ang_n[dst.long()[p]]
    ref3 = torch.cat([torch.bmm(attr[p].unsqueeze(-2), R).squeeze(-2), torch.cos(dth).unsqueeze(-1),
                      torch.sin(dth).unsqueeze(-1)], dim=-1)
    close(o3, ref3, atol=1e-6)


# ---------------------------------------------------------------------------------------------
# segment attention
# ---------------------------------------------------------------------------------------------
def _random_graph(n, e, gen, empty_rows=True):
    dst = torch.randint(0, n, (e,), generator=gen)
    if empty_rows and n > 3:
        dst[dst == 1] = 0
        dst[dst == n - 1] = 2
    return torch.sort(dst, stable=True).values


@pytest.mark.parametrize("n,e,D,H", [(40, 700, 64, 8), (33, 900, 128, 8), (10, 150, 16, 2), (6, 4000, 128, 8),
                                     (5, 3000, 64, 8), (64, 64, 32, 8), (7, 500, 256, 8), (3, 0, 64, 8)])
def test_seg_attn_forward_backward(n, e, D, H):
    gen = torch.Generator().manual_seed(n * 1000 + e + D)
    dst = _random_graph(n, e, gen)
    q = torch.randn(n, D, generator=gen)
    kv = torch.randn(e, 2 * D, generator=gen)
    if e > 10:
        kv[5, :D] *= 8.0  # a dominant score somewhere: exercises the running-max rescale
    w = torch.randn(n, D, generator=gen)
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    rowptr[1:] = torch.bincount(dst, minlength=n).cumsum(0).int()
    qg, kvg = q.to(DEV).requires_grad_(True), kv.to(DEV).requires_grad_(True)
    out = ops.seg_attn(qg, kvg, rowptr.to(DEV), H)
    (out * w.to(DEV)).sum().backward()
    qc, kvc = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    ref = O._edge_attention(qc[dst], kvc[:, :D], kvc[:, D:], dst, n, H, 0.0, False)
    (ref * w).sum().backward()
    qd, kvd = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    ref64 = O._edge_attention(qd[dst], kvd[:, :D], kvd[:, D:], dst, n, H, 0.0, False)
    (ref64 * w.double()).sum().backward()
    close64(out, ref, ref64, msg="out")
    close64(qg.grad, qc.grad, qd.grad, msg="dq")
    close64(kvg.grad, kvc.grad, kvd.grad, msg="dkv")
    deg = torch.bincount(dst, minlength=n)
    assert torch.all(out.detach().cpu()[deg == 0] == 0)


@pytest.mark.parametrize("n,e,D,H", [(30, 600, 64, 8), (12, 2500, 128, 8)])
def test_seg_attn_with_node_terms(n, e, D, H):
    gen = torch.Generator().manual_seed(e)
    dst = _random_graph(n, e, gen)
    src = torch.randint(0, n, (e,), generator=gen)
    q = torch.randn(n, D, generator=gen)
    kve = torch.randn(e, 2 * D, generator=gen)
    kvn = torch.randn(n, 2 * D, generator=gen)
    w = torch.randn(n, D, generator=gen)
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    rowptr[1:] = torch.bincount(dst, minlength=n).cumsum(0).int()
    src_g = src.int().to(DEV)
    rps, pps = ops.csr_build(src_g, n)
    qg, eg, ng = (t.to(DEV).requires_grad_(True) for t in (q, kve, kvn))
    out = ops.seg_attn(qg, eg, rowptr.to(DEV), H, kv_node=ng, src=src_g, rowptr_by_src=rps, perm_by_src=pps)
    (out * w.to(DEV)).sum().backward()
    qc, ec, nc = (t.clone().requires_grad_(True) for t in (q, kve, kvn))
    key = nc[src][:, :D] + ec[:, :D]
    val = nc[src][:, D:] + ec[:, D:]
    ref = O._edge_attention(qc[dst], key, val, dst, n, H, 0.0, False)
    (ref * w).sum().backward()
    qd, ed, nd_ = (t.double().requires_grad_(True) for t in (q, kve, kvn))
    ref64 = O._edge_attention(qd[dst], nd_[src][:, :D] + ed[:, :D], nd_[src][:, D:] + ed[:, D:], dst, n, H, 0.0, False)
    (ref64 * w.double()).sum().backward()
    close64(out, ref, ref64, msg="out")
    close64(qg.grad, qc.grad, qd.grad, msg="dq")
    close64(eg.grad, ec.grad, ed.grad, msg="dkv_edge")
    close64(ng.grad, nc.grad, nd_.grad, msg="dkv_node")


def test_seg_attn_properties_large():
    """Size-independent properties at the full C3 agent-agent size (~1M edges, D=128)."""
    gen = torch.Generator().manual_seed(11)
    n, e, D, H = 51_200, 972_800, 128, 8
    dst = torch.sort(torch.randint(0, n, (e,), generator=gen)).values
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    deg = torch.bincount(dst, minlength=n)
    rowptr[1:] = deg.cumsum(0).int()
    rp = rowptr.to(DEV)
    q = torch.randn(n, D, device=DEV)
    kv = torch.randn(e, 2 * D, device=DEV)
    # (1) V == 1 -> every non-empty row sums its softmax weights to 1 (minus the 1e-16 guard)
    kv1 = kv.clone()
    kv1[:, D:] = 1.0
    out1 = ops.seg_attn(q, kv1, rp, H)
    nz = (deg > 0).to(DEV)
    assert torch.allclose(out1[nz], torch.ones_like(out1[nz]), rtol=0, atol=2e-6)
    assert torch.all(out1[~nz] == 0)
    # (2) linear in V
    out_a = ops.seg_attn(q, kv, rp, H)
    kv2 = kv.clone()
    kv2[:, D:] *= -2.5
    out_b = ops.seg_attn(q, kv2, rp, H)
    assert torch.allclose(out_b, -2.5 * out_a, rtol=1e-5, atol=1e-6)
    # (3) invariant to a constant added to all keys' score: k -> k + c*q_dir is not constant, so use
    #     the shift invariance through Q scaling to zero: q = 0 gives the plain mean of V over the row
    out_m = ops.seg_attn(torch.zeros_like(q), kv, rp, H)
    mean_v = torch.zeros(n, D, device=DEV).index_add_(0, dst.to(DEV), kv[:, D:]) / deg.clamp(min=1).to(DEV).unsqueeze(-1)
    assert torch.allclose(out_m, mean_v, rtol=1e-4, atol=1e-5)
    # (4) deterministic
    assert torch.equal(ops.seg_attn(q, kv, rp, H), out_a)


def test_seg_attn_dropout_statistics_and_backward_mask():
    gen = torch.Generator().manual_seed(2)
    n, e, D, H, p = 200, 40_000, 64, 8, 0.1
    dst = torch.sort(torch.randint(0, n, (e,), generator=gen)).values
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    rowptr[1:] = torch.bincount(dst, minlength=n).cumsum(0).int()
    q = torch.zeros(n, D, device=DEV)  # uniform attention: out = mean over kept edges / (1-p)
    kv = torch.ones(e, 2 * D, device=DEV).requires_grad_(True)
    torch.manual_seed(0)
    out = ops.seg_attn(q, kv, rowptr.to(DEV), H, dropout_p=p, training=True)
    # E[out] = 1; per (row, head) variance ~ p/(1-p)/deg  (deg ~ 200)
    assert abs(out.mean().item() - 1.0) < 5e-3
    assert 0.01 < out.std().item() < 0.05
    out.sum().backward()
    dv = kv.grad[:, D:].view(e, H, D // H)
    # dV = alpha * mask/(1-p) * dOut: zero exactly where the forward dropped the edge
    dropped = (dv == 0).all(dim=-1).float().mean().item()
    assert abs(dropped - p) < 0.01
    # eval mode ignores dropout
    out_eval = ops.seg_attn(q, kv.detach(), rowptr.to(DEV), H, dropout_p=p, training=False)
    assert torch.allclose(out_eval, torch.ones_like(out_eval), atol=2e-6)


def test_dropout_seed_from_device_memory():
    # a device-resident seed is added to the per-call constant: same masks as the summed host seed,
    # forward and backward, and different masks once the device value moves
    gen = torch.Generator().manual_seed(4)
    n, e, D, H, p = 50, 4000, 64, 8, 0.25
    dst = torch.sort(torch.randint(0, n, (e,), generator=gen)).values
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    rowptr[1:] = torch.bincount(dst, minlength=n).cumsum(0).int()
    rowptr = rowptr.to(DEV)
    q = torch.randn(n, D, generator=gen).to(DEV).requires_grad_(True)
    kv = torch.randn(e, 2 * D, generator=gen).to(DEV).requires_grad_(True)
    go = torch.randn(n, D, generator=gen).to(DEV)

    def run(device_seed):
        ops.DEVICE_SEED, ops._seed_calls = device_seed, 0
        try:
            out = ops.seg_attn(q, kv, rowptr, H, dropout_p=p, training=True)
            return (out,) + torch.autograd.grad(out, (q, kv), go)
        finally:
            ops.DEVICE_SEED = None

    seed = torch.tensor([123456789], dtype=torch.int64, device=DEV)
    a = run(seed)
    const = 0x9E3779B97F4A7C15 & ((1 << 62) - 1)  # constant of the first call after the counter reset
    out = ops._SegAttn.apply(q, kv, None, rowptr, None, None, None, H, p, const + 123456789, "", False)
    b = (out,) + torch.autograd.grad(out, (q, kv), go)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    seed.add_(1)
    c = run(seed)
    assert not torch.equal(a[0], c[0])

    # temporal attention takes the same device seed
    S, N = 21, 40
    qkv = torch.randn(S, N, 3 * D, generator=gen).to(DEV).requires_grad_(True)
    g2 = torch.randn(S, N, D, generator=gen).to(DEV)

    def run_t(device_seed):
        ops.DEVICE_SEED, ops._seed_calls = device_seed, 0
        try:
            o = ops.temporal_attention(qkv, H, p, True)
            return (o,) + torch.autograd.grad(o, (qkv,), g2)
        finally:
            ops.DEVICE_SEED = None

    t1 = run_t(seed)
    o = ops._TemporalAttn.apply(qkv, H, p, const + int(seed.item()))
    t2 = (o,) + torch.autograd.grad(o, (qkv,), g2)
    for x, y in zip(t1, t2):
        assert torch.equal(x, y)


def test_bad_arguments_raise():
    q = torch.zeros(4, 24, device=DEV)
    kv = torch.zeros(3, 48, device=DEV)
    rp = torch.tensor([0, 1, 2, 3, 3], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="embed_dim"):
        ops.seg_attn(q, kv, rp, 2)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.seg_attn(torch.zeros(4, 64), torch.zeros(3, 128), rp.cpu(), 8)


# ---------------------------------------------------------------------------------------------
# gated update, fused loss
# ---------------------------------------------------------------------------------------------
def test_gated_update():
    gen = torch.Generator().manual_seed(4)
    n, D = 333, 128
    agg, ih, hs = torch.randn(n, D, generator=gen), torch.randn(n, D, generator=gen), torch.randn(n, 2 * D, generator=gen)
    w = torch.randn(n, D, generator=gen)
    g = [t.to(DEV).requires_grad_(True) for t in (agg, ih, hs)]
    out = ops.gated_update(*g)
    (out * w.to(DEV)).sum().backward()
    c = [t.clone().requires_grad_(True) for t in (agg, ih, hs)]
    ref = c[0] + torch.sigmoid(c[1] + c[2][:, :D]) * (c[2][:, D:] - c[0])
    (ref * w).sum().backward()
    close(out, ref)
    for a, b in zip(g, c):
        close(a.grad, b.grad)


@pytest.mark.parametrize("scenes", [1, 6])
def test_fused_supervised_loss_vs_oracle(scenes):
    b = hv.make_batch(scenes, first_index=40)
    n = int(b.num_nodes)
    gen = torch.Generator().manual_seed(scenes)
    y_hat = torch.randn(6, n, 30, 4, generator=gen)
    y_hat[..., :2] = b["y"].unsqueeze(0) + 2.0 * torch.randn(6, n, 30, 2, generator=gen)
    y_hat[..., 2:] *= 5.0
    y_hat[0, 0, 0, 2], y_hat[1, 1, 1, 3] = 12.0, -11.0
    pi = torch.randn(n, 6, generator=gen)
    yc, pc = y_hat.clone().requires_grad_(True), pi.clone().requires_grad_(True)
    reg_c, cls_c, best_c = O.supervised_losses(yc, pc, b["y"], b["padding_mask"], 20)
    (1.3 * reg_c + 0.7 * cls_c).backward()
    yg, pg = y_hat.to(DEV).requires_grad_(True), pi.to(DEV).requires_grad_(True)
    reg, cls, best = ops.supervised_loss(yg, pg, b["y"].to(DEV), b["padding_mask"].to(DEV), 20)
    (1.3 * reg + 0.7 * cls).backward()
    close(reg, reg_c)
    close(cls, cls_c)
    valid = (~b["padding_mask"][:, 20:]).any(dim=1)
    assert torch.equal(best.cpu()[valid], best_c[valid])
    close(yg.grad, yc.grad, atol=1e-7)
    close(pg.grad, pc.grad, atol=1e-7)


def test_fused_supervised_loss_all_padded_is_zero():
    n = 5
    pm = torch.ones(n, 50, dtype=torch.bool, device=DEV)
    y_hat = torch.randn(6, n, 30, 4, device=DEV, requires_grad=True)
    pi = torch.randn(n, 6, device=DEV, requires_grad=True)
    reg, cls, best = ops.supervised_loss(y_hat, pi, torch.zeros(n, 30, 2, device=DEV), pm, 20)
    (reg + cls).backward()
    assert reg.item() == 0.0 and cls.item() == 0.0
    assert torch.all(y_hat.grad == 0) and torch.all(pi.grad == 0) and torch.all(best == 0)


# ---------------------------------------------------------------------------------------------
# dense projections on the fp32 matrix cores
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N,ln,relu,use_add", [
    (1000, 128, 128, True, True, False), (777, 256, 128, True, True, True), (300, 128, 256, False, False, False),
    (515, 128, 512, False, True, False), (64, 512, 128, False, False, False), (2500, 64, 64, True, False, False),
    (130, 2, 128, True, True, False), (129, 16, 16, True, True, True), (50, 128, 60, False, False, False),
    (1, 128, 128, True, False, False), (4099, 128, 768, False, False, False), (333, 32, 32, False, True, True)])
def test_fused_linear_forward_backward(M, K, N, ln, relu, use_add):
    gen = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=gen)
    w = torch.randn(N, K, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen)
    add = torch.randn(M, N, generator=gen) if use_add else None
    norm = torch.nn.LayerNorm(N) if ln else None
    if ln:
        with torch.no_grad():
            norm.weight.copy_(1 + 0.3 * torch.randn(N, generator=gen))
            norm.bias.copy_(0.2 * torch.randn(N, generator=gen))
    gy = torch.randn(M, N, generator=gen)

    def run(dtype):
        xs, ws, bs = (t.to(dtype).clone().requires_grad_(True) for t in (x, w, b))
        ad = add.to(dtype).clone().requires_grad_(True) if use_add else None
        y = torch.nn.functional.linear(xs, ws, bs)
        if ad is not None:
            y = y + ad
        g = be = None
        if ln:
            g = norm.weight.detach().to(dtype).clone().requires_grad_(True)
            be = norm.bias.detach().to(dtype).clone().requires_grad_(True)
            y = torch.nn.functional.layer_norm(y, (N,), g, be, 1e-5)
        if relu:
            y = torch.relu(y)
        (y * gy.to(dtype)).sum().backward()
        return y, [xs.grad, ws.grad, bs.grad, ad.grad if ad is not None else None, g.grad if ln else None,
                   be.grad if ln else None]

    y32, g32 = run(torch.float32)
    y64, g64 = run(torch.float64)
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    ag = add.to(DEV).requires_grad_(True) if use_add else None
    norm_g = None
    if ln:
        norm_g = torch.nn.LayerNorm(N).to(DEV)
        norm_g.load_state_dict(norm.state_dict())
    y = ops.fused_linear(xg, wg, bg, add=ag, layer_norm=norm_g, relu=relu)
    (y * gy.to(DEV)).sum().backward()
    close64(y, y32, y64, msg="y")
    grads = [xg.grad, wg.grad, bg.grad, ag.grad if use_add else None, norm_g.weight.grad if ln else None,
             norm_g.bias.grad if ln else None]
    for name, a, r32, r64 in zip(["dx", "dw", "db", "dadd", "dgamma", "dbeta"], grads, g32, g64):
        if r64 is not None:
            close64(a, r32, r64, msg=name)


def test_fused_linear_strided_input_and_3d():
    gen = torch.Generator().manual_seed(8)
    feat = torch.randn(500, 4, generator=gen)
    w = torch.randn(64, 2, generator=gen)
    b = torch.randn(64, generator=gen)
    fg = feat.to(DEV)
    y = ops.fused_linear(fg[:, 2:4], w.to(DEV), b.to(DEV))
    close(y, torch.nn.functional.linear(feat[:, 2:4], w, b))
    x3 = torch.randn(20, 33, 64, generator=gen)
    w3 = torch.randn(64, 64, generator=gen) / 8
    y3 = ops.fused_linear(x3.to(DEV), w3.to(DEV), None, relu=True)
    assert y3.shape == (20, 33, 64)
    close(y3, torch.relu(torch.nn.functional.linear(x3, w3)))


# ---------------------------------------------------------------------------------------------
# temporal encoder attention core
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,N,D,H", [(21, 37, 128, 8), (21, 50, 64, 8), (5, 9, 16, 2), (32, 3, 64, 8), (1, 4, 128, 8)])
def test_temporal_attention_vs_oracle(S, N, D, H):
    gen = torch.Generator().manual_seed(S * N)
    qkv = torch.randn(S, N, 3 * D, generator=gen)
    w = torch.randn(S, N, D, generator=gen)

    def ref(dtype):
        t = qkv.to(dtype).clone().requires_grad_(True)
        d = D // H
        q, k, v = (t[..., i * D:(i + 1) * D].reshape(S, N * H, d).transpose(0, 1) for i in range(3))
        mask = torch.triu(torch.full((S, S), float("-inf"), dtype=dtype), diagonal=1)
        p = torch.softmax(torch.baddbmm(mask, q / d ** 0.5, k.transpose(-2, -1)), dim=-1)
        o = torch.bmm(p, v).transpose(0, 1).reshape(S, N, D)
        (o * w.to(dtype)).sum().backward()
        return o, t.grad

    o32, g32 = ref(torch.float32)
    o64, g64 = ref(torch.float64)
    tg = qkv.to(DEV).requires_grad_(True)
    o = ops.temporal_attention(tg, H)
    (o * w.to(DEV)).sum().backward()
    close64(o, o32, o64, msg="out")
    close64(tg.grad, g32, g64, msg="d_qkv")


@pytest.mark.parametrize("S,N", [(21, 37), (32, 5), (1, 4), (7, 300)])
def test_temporal_attention_on_the_matrix_cores_vs_oracle(S, N):
    """D = 128 with 16-wide heads while the dense products run on bfloat16 operands: the attention core takes the
    matrix-core kernels.  Forward products and the scores / dP of the backward take split operands (sixteen mantissa
    bits): outputs within 1e-4 of the float64 oracle.  The three gradient products take operands rounded to bfloat16
    like the dense layers' gradient products of this mode: gradients within 5e-3 normwise and 2e-2 of the tensor's
    scale elementwise (the mode's bar for gradients).  With dropout the kernels draw the very masks of the scalar
    kernels, so both routes agree to the same bars under the same seed."""
    D, H = 128, 8
    gen = torch.Generator().manual_seed(S * N + 1)
    qkv = torch.randn(S, N, 3 * D, generator=gen)
    w = torch.randn(S, N, D, generator=gen)
    t = qkv.double().clone().requires_grad_(True)
    d = D // H
    q, k, v = (t[..., i * D:(i + 1) * D].reshape(S, N * H, d).transpose(0, 1) for i in range(3))
    mask = torch.triu(torch.full((S, S), float("-inf"), dtype=torch.float64), diagonal=1)
    prob = torch.softmax(torch.baddbmm(mask, q / d ** 0.5, k.transpose(-2, -1)), dim=-1)
    o64 = torch.bmm(prob, v).transpose(0, 1).reshape(S, N, D)
    (o64 * w.double()).sum().backward()

    def run(mode, p=0.0):
        previous = ops.set_matmul_precision(mode)
        try:
            tg = qkv.to(DEV).requires_grad_(True)
            torch.manual_seed(5)
            with ops.seed_scope():  # (draws its base from the host generator on entry)
                o = ops.temporal_attention(tg, H, dropout_p=p, training=p > 0.0)
                (o * w.to(DEV)).sum().backward()
            return o.detach(), tg.grad
        finally:
            ops.set_matmul_precision(previous)

    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    o, g = run("bf16")
    for name, got, want, normwise, elementwise in (("out", o, o64, 1e-4, 1e-4), ("d_qkv", g, t.grad, 5e-3, 2e-2)):
        e = _parity.record(test, name, got, want, normwise)
        assert e <= normwise, (name, e)
        scale = float(want.detach().abs().max())
        assert float((got.cpu().double() - want.detach()).abs().max()) <= elementwise * scale, name
    if S > 1:
        o_scalar, g_scalar = run("fp32", 0.3)
        o_drop, g_drop = run("bf16", 0.3)
        assert float((o_drop - o_scalar).abs().max()) <= 1e-4 * float(o_scalar.abs().max())
        assert float((g_drop - g_scalar).abs().max()) <= 2e-2 * float(g_scalar.abs().max())
        assert float((g_drop - g_scalar).norm()) <= 5e-3 * float(g_scalar.norm())
        assert float((o_drop - o).abs().max()) > 1e-2  # (the masks did something)


def test_temporal_attention_dropout_mask_consistency():
    S, N, D, H, p = 21, 200, 128, 8, 0.2
    qkv = torch.zeros(S, N, 3 * D, device=DEV)
    qkv[..., 2 * D:] = 1.0  # uniform attention over the causal prefix, V == 1
    qkv.requires_grad_(True)
    torch.manual_seed(3)
    o = ops.temporal_attention(qkv, H, dropout_p=p, training=True)
    # row i averages i+1 kept/dropped weights scaled by 1/(1-p): mean 1
    assert abs(o.mean().item() - 1.0) < 0.01
    assert o[0].unique().numel() == 2  # first step has a single key: either dropped (0) or 1/(1-p)
    o.sum().backward()
    assert torch.isfinite(qkv.grad).all()
    assert torch.allclose(ops.temporal_attention(qkv.detach(), H, dropout_p=p, training=False),
                          torch.ones(S, N, D, device=DEV), atol=1e-6)


@pytest.mark.parametrize("shape", [(1, 128), (5, 64), (1000, 128), (21, 300, 128), (4097, 100), (333, 7)])
def test_layer_norm_kernel_vs_framework(shape):
    gen = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=gen) * 3 + 1).to(DEV).requires_grad_(True)
    norm = torch.nn.LayerNorm(shape[-1]).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(shape[-1], generator=gen))
        norm.bias.copy_(torch.randn(shape[-1], generator=gen))
    go = torch.randn(*shape, generator=gen).to(DEV)
    y = ops.layer_norm(x, norm)
    gx, gw, gb = torch.autograd.grad(y, (x, norm.weight, norm.bias), go)
    xr = x.detach().double().requires_grad_(True)
    wr, br = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (shape[-1],), wr, br, norm.eps)
    rx, rw, rb = torch.autograd.grad(yr, (xr, wr, br), go.double())
    close(y, yr.float(), 1e-5)
    close(gx, rx.float(), 1e-5)
    close(gw, rw.float(), 2e-5)
    close(gb, rb.float(), 2e-5)
    # a strided view (every other row of a wider buffer) goes through the same kernels
    if len(shape) == 2 and shape[0] > 1:
        wide = torch.randn(shape[0], 2 * shape[1], generator=gen).to(DEV)
        view = wide[:, :shape[1]]
        close(ops.layer_norm(view, norm), norm(view), 1e-5)


@pytest.mark.parametrize("M,D,B", [(1, 128, 2), (7, 64, 1), (1000, 128, 1), (68_617, 128, 2), (4099, 64, 2)])
def test_embed_first_stage_kernel(M, D, B):
    """ReLU(LN(Linear(2, D))) of one or two branches in one launch against the framework's layers,
    forward and every gradient (the kernel recomputes the pre-norm values in backward)."""
    gen = torch.Generator().manual_seed(M + D + B)
    f = (torch.randn(M, 2 * B, generator=gen) * 4).to(DEV).requires_grad_(True)
    lins = [torch.nn.Linear(2, D).to(DEV) for _ in range(B)]
    norms = [torch.nn.LayerNorm(D).to(DEV) for _ in range(B)]
    with torch.no_grad():
        for n_ in norms:
            n_.weight.copy_(torch.randn(D, generator=gen))
            n_.bias.copy_(torch.randn(D, generator=gen) * 0.3)
    go = torch.randn(M, B * D, generator=gen).to(DEV)
    params = [(l.weight, l.bias, n_.weight, n_.bias) for l, n_ in zip(lins, norms)]
    flat = [t for g in params for t in g]
    h = ops.embed2(f, params, norms[0].eps)
    got = torch.autograd.grad(h, [f] + flat, go)

    f64 = f.detach().double().requires_grad_(True)
    flat64 = [t.detach().double().requires_grad_(True) for t in flat]
    outs = []
    for b in range(B):
        w, bias, g, be = flat64[4 * b:4 * b + 4]
        z = f64[:, 2 * b:2 * b + 2] @ w.t() + bias
        outs.append(torch.relu(torch.nn.functional.layer_norm(z, (D,), g, be, norms[0].eps)))
    ref = torch.cat(outs, dim=-1)
    want = torch.autograd.grad(ref, [f64] + flat64, go.double())
    close(h, ref.float(), 1e-5)
    for a, b_ in zip(got, want):
        # ReLU kinks: an element within rounding of zero may flip; bound the comparison by the tensor scale
        close(a, b_.float(), 1e-4, atol=2e-5)


@pytest.mark.parametrize("M,K,N,codes", [(5000, 256, 128, 12), (333, 64, 64, 3), (1, 128, 128, 16)])
def test_fused_linear_with_table_rows_added(M, K, N, codes):
    """A small table whose rows are added per input row (categorical embeddings) inside the LayerNorm
    epilogue; its gradient is the per-code sum of the pre-norm gradient rows."""
    gen = torch.Generator().manual_seed(M + K + codes)
    x = torch.randn(M, K, generator=gen).to(DEV).requires_grad_(True)
    lin = torch.nn.Linear(K, N).to(DEV)
    norm = torch.nn.LayerNorm(N).to(DEV)
    table = torch.randn(codes, N, generator=gen).to(DEV).requires_grad_(True)
    idx = torch.randint(0, codes, (M,), generator=gen).to(torch.int32).to(DEV)
    go = torch.randn(M, N, generator=gen).to(DEV)
    y = ops.fused_linear(x, lin.weight, lin.bias, add=table, layer_norm=norm, relu=True, add_index=idx)
    got = torch.autograd.grad(y, (x, lin.weight, lin.bias, table, norm.weight, norm.bias), go)
    ref = torch.relu(norm(lin(x) + table.index_select(0, idx.long())))
    want = torch.autograd.grad(ref, (x, lin.weight, lin.bias, table, norm.weight, norm.bias), go)
    close(y, ref, 1e-5, atol=1e-5)
    for a, b in zip(got, want):
        close(a, b, 1e-4, atol=2e-5)
    # indices outside the table are clamped on the way in and ignored on the way back
    bad = idx.clone()
    bad[0] = 1000
    y2 = ops.fused_linear(x, lin.weight, lin.bias, add=table, layer_norm=norm, relu=True, add_index=bad)
    fixed = idx.clone()
    fixed[0] = codes - 1
    ref2 = torch.relu(norm(lin(x) + table.index_select(0, fixed.long())))
    close(y2, ref2, 1e-5, atol=1e-5)


@pytest.mark.parametrize("relu", [False, True])
def test_fused_linear_dropout_and_residual_epilogue(relu):
    """y = residual + dropout([relu](x W^T + b)): the kept fraction, the kept values, and gradients that
    use exactly the forward's mask (read back from the output and replayed through the framework)."""
    gen = torch.Generator().manual_seed(31 + relu)
    M, K, N, p = 3001, 128, 256, 0.3
    x = torch.randn(M, K, generator=gen).to(DEV).requires_grad_(True)
    lin = torch.nn.Linear(K, N).to(DEV)
    res = None if relu else torch.randn(M, N, generator=gen).to(DEV).requires_grad_(True)
    go = torch.randn(M, N, generator=gen).to(DEV)
    torch.manual_seed(9)
    y = ops.fused_linear(x, lin.weight, lin.bias, relu=relu, dropout_p=p, training=True, residual=res)
    inputs = (x, lin.weight, lin.bias) + (() if res is None else (res,))
    got = torch.autograd.grad(y, inputs, go)
    with torch.no_grad():
        pre = lin(x)
        pre = torch.relu(pre) if relu else pre
        delta = y if res is None else y - res
        live = pre.abs() > 1e-6  # where the pre-dropout value is not (numerically) zero the mask is visible
        kept = (delta.abs() > 0) & live
        frac = 1.0 - kept[live].float().mean().item()
        assert abs(frac - p) < 0.01, frac
        close(delta[kept], (pre / (1 - p))[kept], 1e-5, atol=1e-5)
        mask = torch.where(live, kept, torch.ones_like(kept)).float() / (1 - p)
    ref = lin(x)
    ref = (torch.relu(ref) if relu else ref) * mask
    if res is not None:
        ref = ref + res
    want = torch.autograd.grad(ref, inputs, go)
    # where the pre-dropout value is numerically zero the output does not show whether the element was dropped;
    # rows of x and rows of W / entries of b that such an element feeds are left out of the comparison
    unsure = ~live & (pre != 0) if relu else ~live  # an exact zero behind the ReLU passes no gradient either way
    rows_ok, cols_ok = ~unsure.any(dim=1), ~unsure.any(dim=0)
    assert int(rows_ok.sum()) > M - 8 and int(cols_ok.sum()) > N - 8
    close(got[0][rows_ok], want[0][rows_ok], 1e-4, atol=2e-5)
    close(got[1][cols_ok], want[1][cols_ok], 1e-4, atol=2e-5)
    close(got[2][cols_ok], want[2][cols_ok], 1e-4, atol=2e-5)
    for a, b in zip(got[3:], want[3:]):
        close(a, b, 1e-4, atol=2e-5)
    # evaluation mode: no dropout
    y_eval = ops.fused_linear(x, lin.weight, lin.bias, relu=relu, dropout_p=p, training=False, residual=res)
    ref_eval = torch.relu(lin(x)) if relu else lin(x) + res
    close(y_eval, ref_eval, 1e-5, atol=1e-5)


@pytest.mark.parametrize("B", [1, 300, 2623])
def test_fused_adversarial_loss_terms_match_framework_ops(B):
    """hivt_hinge_r1_fwd/bwd and hivt_weighted_mean_fwd/bwd against the framework expressions of
    losses/adv_losses.py (masked means with a keep weight per row)."""
    import torch.nn.functional as F
    torch.manual_seed(B)
    dev = torch.device("cuda", 0)
    s_real = (torch.randn(B, device=dev) * 1.5).requires_grad_(True)
    s_fake = (torch.randn(B, device=dev) * 1.5).requires_grad_(True)
    gn2 = torch.rand(B, device=dev).requires_grad_(True)
    w = (torch.rand(B, device=dev) > 0.3).float()
    cnt = w.sum().clamp(min=1.0)
    lam = 0.7
    loss, r1 = ops.hinge_r1_loss(s_real, gn2, s_fake, w, cnt, lam)
    (loss * 1.3).backward()
    got = [t.grad.clone() for t in (s_real, gn2, s_fake)]
    for t in (s_real, gn2, s_fake):
        t.grad = None
    r1_ref = (gn2 * w).sum() / cnt
    ref = (F.relu(1.0 - s_real) * w).sum() / cnt + (F.relu(1.0 + s_fake) * w).sum() / cnt + lam * r1_ref
    (ref * 1.3).backward()
    assert not r1.requires_grad
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert abs(float(r1) - float(r1_ref)) <= 1e-5 * max(1.0, abs(float(r1_ref)))
    for g, t in zip(got, (s_real, gn2, s_fake)):
        assert float((g - t.grad).abs().max()) <= 1e-6 * max(1e-6, float(t.grad.abs().max()))
    for t in (s_real, gn2, s_fake):
        t.grad = None
    g = ops.weighted_mean(s_fake, w, cnt, -1.0)
    (g * 0.4).backward()
    got = s_fake.grad.clone()
    s_fake.grad = None
    ref = -(s_fake * w).sum() / cnt
    (ref * 0.4).backward()
    assert abs(float(g) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert float((got - s_fake.grad).abs().max()) <= 1e-6 * max(1e-6, float(s_fake.grad.abs().max()))


@pytest.mark.parametrize("shape", [(300, 128), (21, 50, 64), (5, 100)])
def test_layer_norm_residual_sums_both_gradient_routes(shape):
    """x + f(norm(x)) written with ``layer_norm_residual``: outputs and the gradient of x (residual route
    plus LayerNorm route, summed inside the LayerNorm-backward kernel) against the framework ops."""
    torch.manual_seed(shape[-1])
    dev = torch.device("cuda", 0)
    norm = torch.nn.LayerNorm(shape[-1]).to(dev)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.2, 0.2)
    x = torch.randn(*shape, device=dev, requires_grad=True)
    gy = torch.randn(*shape, device=dev)
    normed, again = ops.layer_norm_residual(x, norm)
    out = again * 0.5 + torch.tanh(normed)
    out.backward(gy)
    got = [x.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()]
    x.grad = norm.weight.grad = norm.bias.grad = None
    ref = x * 0.5 + torch.tanh(norm(x))
    ref.backward(gy)
    assert float((out - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    for g, r in zip(got, (x.grad, norm.weight.grad, norm.bias.grad)):
        assert float((g - r).abs().max()) <= 2e-5 * float(r.abs().max())
    # either output alone
    x.grad = None
    ops.layer_norm_residual(x, norm)[1].mul(2.0).backward(gy)
    assert torch.equal(x.grad, 2.0 * gy)


@pytest.mark.parametrize("M,D,p", [(700, 128, 0.0), (2623, 64, 0.0), (3000, 128, 0.1)])
def test_pre_norm_feed_forward_block(M, D, p):
    """x + drop(lin2(drop(relu(lin1(norm(x)))))) as one autograd node: output and all seven gradients
    against the framework ops; with dropout active, against the framework ops fed the kernel's own masks
    (read back from the block's saved activations)."""
    torch.manual_seed(M + D)
    dev = torch.device("cuda", 0)
    norm = torch.nn.LayerNorm(D).to(dev)
    lin1, lin2 = torch.nn.Linear(D, 4 * D).to(dev), torch.nn.Linear(4 * D, D).to(dev)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.2, 0.2)
    params = [norm.weight, norm.bias, lin1.weight, lin1.bias, lin2.weight, lin2.bias]
    x = torch.randn(M, D, device=dev, requires_grad=True)
    gy = torch.randn(M, D, device=dev)
    assert ops.pre_norm_feed_forward_supported(x, lin1, lin2)
    torch.manual_seed(99)
    y = ops.pre_norm_feed_forward(x, norm, lin1, p, lin2, p, True)
    h_saved = y.grad_fn.saved_tensors[5].clone()  # hidden activation kept by the node
    y.backward(gy)
    got = [x.grad.clone()] + [q.grad.clone() for q in params]
    x.grad = None
    for q in params:
        q.grad = None
    z1 = lin1(norm(x))
    if p == 0.0:
        hidden = torch.relu(z1)
        ref = x + lin2(hidden)
    else:
        # masks of the kernel run: from the saved hidden activation, and from the output
        # active-and-kept units are the positive entries of the saved activation (the ReLU decision is
        # taken from the kernel too: a pre-activation within rounding of zero may fall on either side)
        hidden = z1 * (h_saved > 0) / (1.0 - p)
        pre = lin2(hidden)
        keep2 = ((y.detach() - x.detach()).abs() > 0) | (pre.detach() == 0)
        frac = 1.0 - float(keep2.float().mean())
        assert abs(frac - p) < 0.01 and abs(1.0 - float(((h_saved > 0) | (z1.detach() <= 0)).float().mean()) - p / 2) < 0.02
        ref = x + pre * keep2 / (1.0 - p)
    ref.backward(gy)
    assert float((y - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    for g, r, name in zip(got, [x.grad] + [q.grad for q in params], ["x", "gamma", "beta", "w1", "b1", "w2", "b2"]):
        assert float((g - r).abs().max()) <= 3e-5 * float(r.abs().max()), name


@pytest.mark.parametrize("N,D", [(900, 128), (2623, 64)])
def test_gated_update_tail_block(N, D):
    """Gate + update + out_proj + residual as one autograd node against the framework ops
    (local_encoder.py:183-188,197-199), all eleven gradients."""
    torch.manual_seed(N)
    dev = torch.device("cuda", 0)
    lin_ih, lin_hh, lin_self, out_proj = (torch.nn.Linear(D, D).to(dev) for _ in range(4))
    agg = torch.randn(N, D, device=dev, requires_grad=True)
    c = torch.randn(N, D, device=dev, requires_grad=True)
    res = torch.randn(N, D, device=dev, requires_grad=True)
    gy = torch.randn(N, D, device=dev)
    leaves = [agg, c, res] + [p for m in (lin_ih, lin_hh, lin_self, out_proj) for p in (m.weight, m.bias)]
    y = ops.gated_update_tail(agg, c, res, lin_ih, lin_hh, lin_self, out_proj, 0.1, False)
    y.backward(gy)
    got = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None
    gate = torch.sigmoid(lin_ih(agg) + lin_hh(c))
    ref = res + out_proj(agg + gate * (lin_self(c) - agg))
    ref.backward(gy)
    assert float((y - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    for g, t in zip(got, leaves):
        assert float((g - t.grad).abs().max()) <= 3e-5 * float(t.grad.abs().max())
    # with dropout: kept fraction and forward / backward mask agreement through the residual-free part
    torch.manual_seed(1)
    y = ops.gated_update_tail(agg, c, res, lin_ih, lin_hh, lin_self, out_proj, 0.25, True)
    kept = ((y.detach() - res.detach()).abs() > 0).float()
    assert abs(float(kept.mean()) - 0.75) < 0.01
    for t in leaves:
        t.grad = None
    y.backward(gy)
    assert torch.equal(res.grad, gy)
    ref_d = res + kept / 0.75 * out_proj(agg + gate.detach() * 0 + torch.sigmoid(lin_ih(agg) + lin_hh(c)) * (lin_self(c) - agg))
    for t in leaves:
        t.grad = None
    ref_d.backward(gy)
    y2_err = float((y.detach() - ref_d.detach()).abs().max())
    assert y2_err <= 3e-5 * float(ref_d.abs().max())


# ---------------------------------------------------------------------------------------------
# dense kernels at the training step's largest sizes (grids of more than a thousand tiles)
# ---------------------------------------------------------------------------------------------
def _close_scale(a, b, rtol, msg):
    err = float((a.double() - b.double()).abs().max())
    scale = max(float(b.double().abs().max()), 1e-30)
    assert err <= rtol * scale, f"{msg}: {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("M,K,N,relu,use_add,use_res", [
    (70_001, 128, 128, False, False, False), (66_000, 64, 128, True, False, False), (40_003, 128, 256, True, True, False),
    (33_000, 256, 384, False, False, True), (70_000, 512, 128, False, False, True), (136_000, 128, 128, False, True, False)])
def test_linear_kernels_large_grids(M, K, N, relu, use_add, use_res):
    """Forward (bias, added rows, ReLU, residual) and the data / weight gradients at the sizes of the
    full training step, against float64 framework ops on the same device."""
    torch.manual_seed(M % 1000 + K + N)
    x = torch.randn(M, K, device=DEV, requires_grad=True)
    w = (torch.randn(N, K, device=DEV) / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, device=DEV, requires_grad=True)
    add = torch.randn(M, N, device=DEV, requires_grad=True) if use_add else None
    res = torch.randn(M, N, device=DEV, requires_grad=True) if use_res else None
    gy = torch.randn(M, N, device=DEV)
    y = ops.fused_linear(x, w, b, add=add, relu=relu, residual=res)
    y.backward(gy)
    leaves = [t for t in (x, w, b, add, res) if t is not None]
    got = [t.grad.clone() for t in leaves]
    d = [t.detach().double().requires_grad_(True) for t in leaves]
    it = iter(d)
    xd, wd, bd = next(it), next(it), next(it)
    ref = torch.nn.functional.linear(xd, wd, bd)
    if use_add:
        ref = ref + next(it)
    if relu:  # the on/off decision of units within rounding of zero is taken from the kernel's output
        ref = ref * (y.detach() > 0)
    if use_res:
        ref = ref + next(it)
    ref.backward(gy.double())
    _close_scale(y, ref, 1e-5, "y")
    for name, g, t in zip(["dx", "dw", "db", "d4", "d5"], got, d):
        _close_scale(g, t.grad, 2e-5, name)


def test_linear_dropout_and_blocks_large_grids():
    """Dropout + residual epilogue (kept fraction, forward / backward mask agreement) and the two block
    nodes whose data-gradient products use the gating mask and the added second route, at the row counts
    of the full training step."""
    torch.manual_seed(11)
    M, D, p = 70_000, 128, 0.2
    x = torch.randn(M, D, device=DEV, requires_grad=True)
    lin = torch.nn.Linear(D, D).to(DEV)
    res = torch.randn(M, D, device=DEV)
    gy = torch.randn(M, D, device=DEV)
    y = ops.fused_linear(x, lin.weight, lin.bias, dropout_p=p, training=True, residual=res)
    kept = (y.detach() - res) != 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.005
    y.backward(gy)
    pre = torch.nn.functional.linear(x.detach().double(), lin.weight.detach().double(), lin.bias.detach().double())
    _close_scale(y, res.double() + pre * kept / (1 - p), 1e-5, "y with dropout")
    dx_ref = (gy.double() * kept / (1 - p)) @ lin.weight.detach().double()
    _close_scale(x.grad, dx_ref, 2e-5, "dx with dropout")

    # feed-forward block: the gating mask rides in the second Linear's data-gradient product
    norm = torch.nn.LayerNorm(D).to(DEV)
    lin1, lin2 = torch.nn.Linear(D, 4 * D).to(DEV), torch.nn.Linear(4 * D, D).to(DEV)
    x2 = torch.randn(M, D, device=DEV, requires_grad=True)
    out = ops.pre_norm_feed_forward(x2, norm, lin1, 0.0, lin2, 0.0, False)
    out.backward(gy)
    got = [x2.grad.clone(), lin1.weight.grad.clone(), lin2.weight.grad.clone(), norm.weight.grad.clone()]
    x2.grad = lin1.weight.grad = lin2.weight.grad = norm.weight.grad = None
    ref = x2 + lin2(torch.relu(lin1(norm(x2))))
    ref.backward(gy)
    _close_scale(out, ref, 2e-5, "feed-forward block")
    for name, g, r in zip(["dx", "dw1", "dw2", "dgamma"], got, [x2.grad, lin1.weight.grad, lin2.weight.grad, norm.weight.grad]):
        _close_scale(g, r, 1e-4, "feed-forward block " + name)

    # gated update tail: the aggregated messages receive two gradient routes in one product
    lin_ih, lin_hh, lin_self, out_proj = (torch.nn.Linear(D, D).to(DEV) for _ in range(4))
    agg = torch.randn(M, D, device=DEV, requires_grad=True)
    c = torch.randn(M, D, device=DEV, requires_grad=True)
    r = torch.randn(M, D, device=DEV, requires_grad=True)
    out = ops.gated_update_tail(agg, c, r, lin_ih, lin_hh, lin_self, out_proj, 0.0, False)
    out.backward(gy)
    got = [agg.grad.clone(), c.grad.clone()]
    assert r.grad.data_ptr() != gy.data_ptr()  # a leaf residual gets its own copy of the incoming gradient
    agg.grad = c.grad = None                   # r.grad stays: the second backward accumulates into it in place
    gy_before = gy.clone()
    ref = r + out_proj(agg + torch.sigmoid(lin_ih(agg) + lin_hh(c)) * (lin_self(c) - agg))
    ref.backward(gy)
    _close_scale(out, ref, 2e-5, "update tail")
    _close_scale(got[0], agg.grad, 1e-4, "update tail d_agg")
    _close_scale(got[1], c.grad, 1e-4, "update tail d_center")
    assert torch.equal(gy, gy_before) and torch.equal(r.grad, 2 * gy)


def test_unsupported_gpu_shapes_raise_instead_of_falling_back():
    """No quiet hand-over to the framework's operators for GPU tensors outside the built shapes."""
    x = torch.randn(10, 256, device=DEV)
    norm = torch.nn.LayerNorm(256).to(DEV)
    lin = torch.nn.Linear(256, 256).to(DEV)
    with pytest.raises(ops.UnsupportedShapeError):
        ops.layer_norm(x, norm)
    with pytest.raises(ops.UnsupportedShapeError):
        ops.layer_norm_residual(x, norm)
    with pytest.raises(ops.UnsupportedShapeError):
        ops.fused_linear(x, lin.weight, lin.bias, layer_norm=norm)
    layer = hv.TemporalEncoderLayer(embed_dim=64, num_heads=8, dropout=0.0).to(DEV)
    with pytest.raises(ops.UnsupportedShapeError):
        layer._sa_block(torch.randn(40, 3, 64, device=DEV), torch.randn(40, 3, 64, device=DEV))
    assert ops.fused_linear(x, lin.weight, lin.bias).shape == (10, 256)  # the plain product has no width limit


@pytest.mark.parametrize("steps,n,d,h", [(21, 300, 128, 8), (5, 70, 64, 8), (1, 9, 16, 2)])
def test_seg_attn_over_interleaved_rows(steps, n, d, h):
    """Every target attends over its own `steps` rows of a [steps, N, 2D] tensor (row j * N + n): values and
    gradients against the softmax written out with framework ops, and against the same call on the rows
    re-laid in CSR order."""
    gen = torch.Generator().manual_seed(steps * 1000 + n)
    q = torch.randn(n, d, generator=gen).to(DEV).requires_grad_(True)
    kv = torch.randn(steps, n, 2 * d, generator=gen).to(DEV).requires_grad_(True)
    go = torch.randn(n, d, generator=gen).to(DEV)
    rowptr = (torch.arange(n + 1) * steps).to(torch.int32).to(DEV)
    out = ops.seg_attn(q, kv.view(steps * n, 2 * d), rowptr, h, interleaved=True)
    got = torch.autograd.grad(out, (q, kv), go)
    hd = d // h
    k, v = kv[..., :d].reshape(steps, n, h, hd), kv[..., d:].reshape(steps, n, h, hd)
    scores = (k * q.view(1, n, h, hd)).sum(-1) * hd ** -0.5
    ref = (torch.softmax(scores, dim=0).unsqueeze(-1) * v).sum(0).reshape(n, d)
    want = torch.autograd.grad(ref, (q, kv), go)
    close(out, ref, 1e-5, atol=1e-6)
    for a, b in zip(got, want):
        close(a, b, 1e-5, atol=1e-6)
    # the same edges in CSR order (target-major rows): identical sums, so identical bits
    kv_csr = kv.detach().transpose(0, 1).reshape(n * steps, 2 * d).contiguous().requires_grad_(True)
    out_csr = ops.seg_attn(q, kv_csr, rowptr, h)
    assert torch.equal(out_csr, out)
    g_csr = torch.autograd.grad(out_csr, kv_csr, go)[0].view(n, steps, 2 * d).transpose(0, 1)
    assert torch.equal(g_csr, got[1])
    with pytest.raises(ValueError):
        ops.seg_attn(q, kv.view(steps * n, 2 * d)[:-1], rowptr, h, interleaved=True)


@pytest.mark.parametrize("m,rows,D", [(2500, 12, 128), (46_000, 2600, 128), (700, 30, 64), (1, 1, 128), (300, 500, 64)])
def test_pair_linear_adds_table_rows(m, rows, D):
    """[a(x) | b(x)] + table[index]: the per-source-node key | value rows of the interaction module enter in the
    store epilogue of the edge projection (global_interactor.py:115-119); values and every gradient against the
    stock ops, the table's gradient summed in a fixed order (repeatable bit for bit)."""
    gen = torch.Generator().manual_seed(m + rows)
    x = torch.randn(m, D, generator=gen)
    a, b = torch.nn.Linear(D, D), torch.nn.Linear(D, D)
    table = torch.randn(rows, 2 * D, generator=gen)
    index = torch.randint(0, rows, (m,), generator=gen)
    if rows > 3:
        index[index == 1] = 0  # a table row nobody picks: its gradient is exactly zero
    w = torch.randn(m, 2 * D, generator=gen)
    ag, bg = torch.nn.Linear(D, D).to(DEV), torch.nn.Linear(D, D).to(DEV)
    ag.load_state_dict(a.state_dict())
    bg.load_state_dict(b.state_dict())
    idx_g = index.int().to(DEV)
    rp, pp = ops.csr_build(idx_g, rows)
    xg, tg = x.to(DEV).requires_grad_(True), table.to(DEV).requires_grad_(True)
    y = ops.pair_linear(xg, ag, bg, table=tg, index=idx_g, rowptr_by_index=rp, perm_by_index=pp)
    (y * w.to(DEV)).sum().backward()
    xd, td = x.double().requires_grad_(True), table.double().requires_grad_(True)
    ad, bd = torch.nn.Linear(D, D).double(), torch.nn.Linear(D, D).double()
    ad.load_state_dict(a.state_dict())
    bd.load_state_dict(b.state_dict())
    ref = torch.cat([ad(xd), bd(xd)], dim=-1) + td[index]
    (ref * w.double()).sum().backward()
    close(y, ref, msg="y")
    close(xg.grad, xd.grad, msg="dx")
    close(tg.grad, td.grad, msg="dtable")
    for name, got, want in (("dw_a", ag.weight.grad, ad.weight.grad), ("db_a", ag.bias.grad, ad.bias.grad),
                            ("dw_b", bg.weight.grad, bd.weight.grad), ("db_b", bg.bias.grad, bd.bias.grad)):
        close(got, want, rtol=2e-5, msg=name)
    if rows > 3:
        assert torch.all(tg.grad[1] == 0)
    first = tg.grad.clone()
    tg.grad = None
    y2 = ops.pair_linear(xg, ag, bg, table=tg, index=idx_g, rowptr_by_index=rp, perm_by_index=pp)
    (y2 * w.to(DEV)).sum().backward()
    assert torch.equal(y2, y) and torch.equal(tg.grad, first)
    with pytest.raises(ValueError):
        ops.pair_linear(xg, ag, bg, table=tg, index=idx_g)  # a table that needs a gradient needs the grouping
    with pytest.raises(ValueError):
        ops.pair_linear(xg, ag, bg, table=tg.detach(), index=idx_g.long())


@pytest.mark.parametrize("M,N,R", [(70_001, 128, 32), (5000, 128, 20), (333, 64, 3), (1, 128, 16), (4097, 60, 7),
                                   (141_000, 128, 12)])
def test_rowsum_by_code_matches_index_add_and_repeats(M, N, R):
    """Per-code sums of rows (gradient of a small table added per row): against a float64 index_add, with codes
    outside the table ignored, on a strided input, and bit-identical from call to call (fixed summation order)."""
    gen = torch.Generator().manual_seed(M + N + R)
    wide = torch.randn(M, N + 4, generator=gen).to(DEV)
    x = wide[:, :N]  # row stride N + 4
    idx = torch.randint(0, R, (M,), generator=gen).to(torch.int32)
    if M > 10:
        idx[3], idx[7] = R + 5, -2  # outside the table: contribute nothing
    idx = idx.to(DEV)
    got = ops._rowsum_by_code(x, idx, R)
    keep = (idx >= 0) & (idx < R)
    want = torch.zeros(R, N, dtype=torch.float64, device=DEV).index_add_(0, idx[keep].long(), x[keep].double())
    scale = float(x.abs().sum(0).max()) + 1e-6
    assert float((got.double() - want).abs().max()) < 2e-6 * scale
    assert torch.equal(ops._rowsum_by_code(x, idx, R), got)
