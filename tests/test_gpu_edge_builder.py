"""Device-side all-pairs edge builder (``hivt_edges_all_pairs_csr``): the edge sets it writes straight in
CSR order must equal, bit for bit, the reference's per-timestep filtered lists (CPU oracle) stably sorted by
target -- the order the attention kernels accumulate in -- and what the filter + sort route produces."""
import pytest
import torch

import hivt_gan_amd as hv
from hivt_gan_amd import ops
from hivt_gan_amd.local_encoder import build_edge_plan
from oracle import hivt_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def oracle_csr(b, t0, T, radius):
    """(rowptr, src, dst, attr) of the reference's lists for timesteps t0..t0+T-1, stable-sorted by target."""
    n = int(b.num_nodes)
    data = {k: b[k] for k in b.keys()}
    eis, eas = [], []
    for tt in range(T):
        ei, ea = O.temporal_edges(data, t0 + tt, radius)
        eis.append(ei + tt * n)
        eas.append(ea)
    ei, ea = torch.cat(eis, dim=1), torch.cat(eas, dim=0)
    order = torch.sort(ei[1], stable=True).indices
    deg = torch.bincount(ei[1], minlength=T * n)
    rowptr = torch.cat([deg.new_zeros(1), deg.cumsum(0)])
    return rowptr, ei[0][order], ei[1][order], ea[order]


@pytest.mark.parametrize("scenes,t0,T,radius,kw", [
    (3, 0, 20, 50.0, {}), (8, 0, 20, 20.0, {}), (5, 19, 1, None, {}), (2, 0, 20, None, {}),
    (1, 0, 20, 150.0, dict(num_agents=200, num_lanes=50, all_valid=True, local_radius=150.0)),
])
def test_builder_matches_oracle_lists_sorted_by_target(scenes, t0, T, radius, kw):
    b = hv.make_batch(scenes, **kw)
    assert b["all_pairs"] is True
    g = b.to(DEV)
    n = int(b.num_nodes)
    cap = b["edge_index"].shape[1] * T
    rowptr, src, dst, attr, count = ops.edges_all_pairs_csr(g["scene_range"], g["padding_mask"], g["positions"], t0, T,
                                                            radius, cap)
    rp_ref, src_ref, dst_ref, attr_ref = oracle_csr(b, t0, T, radius)
    e = int(count.item())
    assert e == src_ref.numel()
    assert torch.equal(rowptr.cpu().long(), rp_ref)
    assert torch.equal(src[:e].cpu().long(), src_ref) and torch.equal(dst[:e].cpu().long(), dst_ref)
    assert torch.equal(attr[:e].cpu(), attr_ref)  # same subtraction, bit-exact


def test_both_routes_give_the_same_plan_and_model_outputs():
    torch.manual_seed(3)
    model = hv.HiVT(embed_dim=64, num_heads=8, dropout=0.0, num_temporal_layers=1, num_global_layers=2).to(DEV)
    model.eval()
    b = hv.make_batch(6, first_index=40)
    outs, plans = [], []
    for from_pairs in (False, True):
        g = b.to(DEV)
        plan = build_edge_plan(g, 20, 50.0, from_pairs=from_pairs)
        g["edge_plan"] = plan
        with torch.no_grad():
            outs.append(model(g))
        plans.append(plan)
    old, new = plans
    for a, c in ((old.agent_agent, new.agent_agent), (old.interaction, new.interaction)):
        assert new is not None and c.perm is None and a.perm is not None
        p = a.perm.long()
        assert torch.equal(a.rowptr, c.rowptr)
        assert torch.equal(a.coo_src[p], c.coo_src) and torch.equal(a.coo_dst[p], c.coo_dst)
        assert torch.equal(a.coo_attr[p], c.coo_attr)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_capacity_clamps_the_lists_and_reports_the_full_count():
    b = hv.make_batch(4, first_index=7)
    g = b.to(DEV)
    n = int(b.num_nodes)
    full = ops.edges_all_pairs_csr(g["scene_range"], g["padding_mask"], g["positions"], 0, 20, 50.0,
                                   b["edge_index"].shape[1] * 20)
    e = int(full[4].item())
    cap = e // 2
    rowptr, src, dst, attr, count = ops.edges_all_pairs_csr(g["scene_range"], g["padding_mask"], g["positions"], 0, 20,
                                                            50.0, cap)
    assert int(count.item()) == e and src.shape[0] == cap
    assert torch.equal(rowptr, full[0].clamp(max=cap))
    assert torch.equal(src, full[1][:cap]) and torch.equal(dst, full[2][:cap]) and torch.equal(attr, full[3][:cap])
    assert int(rowptr[-1]) == cap and int(rowptr.max()) == cap


def test_padded_batches_and_non_all_pairs_batches():
    model = hv.HiVT(embed_dim=64, num_heads=8, dropout=0.0, num_temporal_layers=1, num_global_layers=1).to(DEV)
    raw = hv.make_batch(5, first_index=90).to(DEV)
    spec = hv.GanTrainer.calibrate(model, [raw])
    padded = hv.pad_batch(raw, spec)
    assert padded["all_pairs"] is True and padded["scene_range"].shape == (spec.num_nodes, 2)
    plan_raw = build_edge_plan(raw, 20, 50.0)
    plan_pad = build_edge_plan(padded, 20, 50.0)  # fixed capacity, live count on the device
    n_raw, n_pad = int(raw.num_nodes), spec.num_nodes
    e = plan_raw.agent_agent.num_edges
    assert int(plan_pad.agent_agent.count.item()) == e and not bool(plan_pad.overflow)
    # same edges; node ids of timestep t are offset by t * (padded agent count)
    def remap(ids):
        return (ids // n_raw) * n_pad + ids % n_raw
    assert torch.equal(plan_pad.agent_agent.coo_src[:e], remap(plan_raw.agent_agent.coo_src))
    assert torch.equal(plan_pad.agent_agent.coo_dst[:e], remap(plan_raw.agent_agent.coo_dst))
    assert torch.equal(plan_pad.agent_agent.coo_attr[:e], plan_raw.agent_agent.coo_attr)
    # a batch whose candidate list is not all pairs keeps the filter + sort route
    scene = hv.make_scene(3)
    scene["edge_index"] = scene["edge_index"][:, ::2].contiguous()
    odd = hv.collate([scene, hv.make_scene(4)])
    assert odd["all_pairs"] is False
    plan = build_edge_plan(odd.to(DEV), 20, 50.0)
    assert plan.agent_agent.perm is not None
