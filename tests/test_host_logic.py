"""CPU-only checks of the host side: batching offsets, synthetic scene invariants, parameter
naming, the C-ABI surface, and the rounding model the radius filter relies on."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import hivt_gan_amd as hv
from hivt_gan_amd import _lib
from oracle import hivt_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scene_invariants():
    for idx in range(6):
        s = hv.make_scene(idx, random_lane_attrs=bool(idx % 2))
        n, L = int(s.num_nodes), s.lane_vectors.shape[0]
        assert 12 <= n <= 28 and 200 <= L <= 400
        assert s.x.shape == (n, 20, 2) and s.y.shape == (n, 30, 2) and s.positions.shape == (n, 50, 2)
        assert s.padding_mask.dtype == torch.bool and s.bos_mask.shape == (n, 20)
        assert s.edge_index.shape == (2, n * (n - 1)) and s.edge_index.dtype == torch.long
        # source-major permutations order
        assert s.edge_index[0].tolist() == sorted(s.edge_index[0].tolist())
        assert not bool((s.edge_index[0] == s.edge_index[1]).any())
        # no future for agents unseen at the last history step
        assert bool(s.padding_mask[s.padding_mask[:, 19], 20:].all())
        # features are zero on padded steps and at t = 0
        assert float(s.x[:, 0].abs().max()) == 0.0
        assert float(s.y[s.padding_mask[:, 20:]].abs().max() if s.padding_mask[:, 20:].any() else 0.0) == 0.0
        # lane-actor pairs are lane-major, only visible actors, all within the radius
        la = s.lane_actor_index
        assert la.shape[0] == 2 and la.shape[1] == s.lane_actor_vectors.shape[0]
        assert la[0].tolist() == sorted(la[0].tolist())
        assert not bool(s.padding_mask[la[1], 19].any())
        assert bool((torch.norm(s.lane_actor_vectors, dim=1) < 50.0).all())
        assert s.is_intersections.dtype == torch.uint8
    a, b = hv.make_scene(3), hv.make_scene(3)
    assert torch.equal(a.positions, b.positions) and torch.equal(a.lane_actor_index, b.lane_actor_index)


def test_collate_offsets():
    scenes = [hv.make_scene(i) for i in range(3)]
    batch = hv.collate(scenes)
    n0, n1 = int(scenes[0].num_nodes), int(scenes[1].num_nodes)
    l0, l1 = scenes[0].lane_vectors.shape[0], scenes[1].lane_vectors.shape[0]
    assert batch.num_nodes == sum(int(s.num_nodes) for s in scenes) and batch.num_graphs == 3
    e0, e1 = scenes[0].edge_index.shape[1], scenes[1].edge_index.shape[1]
    assert torch.equal(batch.edge_index[:, e0:e0 + e1], scenes[1].edge_index + n0)
    assert torch.equal(batch.edge_index[:, e0 + e1:], scenes[2].edge_index + n0 + n1)
    a0, a1 = scenes[0].lane_actor_index.shape[1], scenes[1].lane_actor_index.shape[1]
    exp = scenes[1].lane_actor_index + torch.tensor([[l0], [n0]])
    assert torch.equal(batch.lane_actor_index[:, a0:a0 + a1], exp)
    assert batch.agent_index.tolist() == [scenes[0].agent_index, scenes[1].agent_index + n0,
                                          scenes[2].agent_index + n0 + n1]
    assert batch.theta.shape == (3,) and batch.origin.shape == (3, 2) and len(batch.seq_id) == 3
    assert torch.equal(batch.x[n0:n0 + n1], scenes[1].x)
    assert batch["padding_mask"] is batch.padding_mask


def test_radius_rounding_model():
    """The device filter computes sqrt(fma(y, y, fl(x*x))) < r; that must be this host's torch.norm."""
    gen = torch.Generator().manual_seed(0)
    a = (torch.rand(300_000, 2, generator=gen) * 2 - 1) * 120
    host = torch.norm(a, p=2, dim=-1).numpy()
    x, y = a[:, 0].numpy(), a[:, 1].numpy()
    model = np.sqrt(((x * x).astype(np.float64) + y.astype(np.float64) * y).astype(np.float32))
    assert np.array_equal(host, model)


def test_oracle_parallel_equals_sequential_and_handles_degenerate_inputs():
    torch.manual_seed(0)
    m = hv.HiVT(embed_dim=16, num_heads=2, dropout=0.0, num_temporal_layers=1, num_global_layers=1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    b = hv.make_batch(2, first_index=9)
    # one agent padded at every history step (no edges, padding tokens), one scene without lanes
    b["padding_mask"][0, :] = True
    b["bos_mask"][0, :] = False
    data = {k: b[k] for k in b.keys()}
    cfg = dict(historical_steps=20, future_steps=30, num_modes=6, rotate=True, num_heads=2, num_temporal_layers=1,
               num_global_layers=1, local_radius=50.0, parallel=False)
    y1, p1, _ = O.hivt_forward(sd, data, cfg)
    y2, p2, _ = O.hivt_forward(sd, data, dict(cfg, parallel=True))
    assert torch.isfinite(y1).all() and torch.allclose(y1, y2, atol=2e-5) and torch.allclose(p1, p2, atol=2e-5)
    data["lane_actor_index"] = torch.zeros(2, 0, dtype=torch.long)
    data["lane_actor_vectors"] = torch.zeros(0, 2)
    y3, _, _ = O.hivt_forward(sd, data, cfg)
    assert torch.isfinite(y3).all()


def test_oracle_segment_softmax_properties():
    gen = torch.Generator().manual_seed(1)
    idx = torch.tensor([0, 0, 0, 2, 2, 5])
    a = torch.randn(6, 4, generator=gen)
    s = O.segment_softmax(a, idx, 7)
    sums = torch.zeros(7, 4).index_add_(0, idx, s)
    assert torch.allclose(sums[[0, 2, 5]], torch.ones(3, 4), atol=1e-6) and torch.all(sums[[1, 3, 4, 6]] == 0)
    perm = torch.tensor([2, 0, 1, 4, 3, 5])
    assert torch.allclose(O.segment_softmax(a[perm], idx[perm], 7), s[perm], atol=1e-7)
    agg = O._edge_attention(torch.randn(6, 8, generator=gen), torch.randn(6, 8, generator=gen),
                            torch.randn(6, 8, generator=gen), idx, 7, 2, 0.0, False)
    assert torch.all(agg[[1, 3, 4, 6]] == 0)


def test_state_dict_keys_match_reference_layout():
    model = hv.HiVT(embed_dim=64, use_gan=True)
    keys = set(model.state_dict().keys())
    for k in ["local_encoder.aa_encoder.center_embed.embed.0.weight",
              "local_encoder.aa_encoder.nbr_embed.module_list.1.3.bias",
              "local_encoder.aa_encoder.nbr_embed.aggr_embed.2.weight", "local_encoder.aa_encoder.bos_token",
              "local_encoder.aa_encoder.lin_ih.weight", "local_encoder.aa_encoder.mlp.3.bias",
              "local_encoder.temporal_encoder.transformer_encoder.layers.3.self_attn.in_proj_weight",
              "local_encoder.temporal_encoder.transformer_encoder.layers.0.self_attn.out_proj.bias",
              "local_encoder.temporal_encoder.transformer_encoder.norm.weight",
              "local_encoder.temporal_encoder.attn_mask", "local_encoder.temporal_encoder.pos_embed",
              "local_encoder.temporal_encoder.padding_token", "local_encoder.temporal_encoder.cls_token",
              "local_encoder.al_encoder.lane_embed.aggr_embed.3.bias", "local_encoder.al_encoder.turn_direction_embed",
              "local_encoder.al_encoder.is_intersection_embed", "local_encoder.al_encoder.traffic_control_embed",
              "global_interactor.rel_embed.module_list.0.0.weight",
              "global_interactor.global_interactor_layers.2.lin_v_edge.weight",
              "global_interactor.global_interactor_layers.0.lin_q_node.bias", "global_interactor.norm.weight",
              "global_interactor.multihead_proj.bias", "decoder.aggr_embed.1.weight", "decoder.loc.3.weight",
              "decoder.pi.6.weight", "decoder.scale.3.bias", "D_long.mlp.9.weight_orig", "D_mid.mlp.6.weight_v",
              "D_short.out.weight_u", "D_short.mlp.3.bias"]:
        assert k in keys, k
    n_gen = sum(p.numel() for p in model.generator_parameters())
    n_cr = sum(p.numel() for p in model.critic_parameters())
    assert (n_gen, n_cr) == (653_369, 374_915)
    big = hv.HiVT(embed_dim=128)
    assert sum(p.numel() for p in big.parameters()) == 2_559_993


def test_reference_loss_classes_match_golden(golden):
    pred = torch.from_numpy(golden["laplace_pred"]).requires_grad_(True)
    val = hv.LaplaceNLLLoss(reduction="mean")(pred, torch.from_numpy(golden["laplace_target"]))
    val.backward()
    np.testing.assert_allclose(val.item(), golden["laplace_mean"], rtol=1e-6)
    np.testing.assert_allclose(pred.grad.numpy(), golden["laplace_mean_grad"], rtol=1e-5, atol=1e-7)
    ce = hv.SoftTargetCrossEntropyLoss(reduction="none")(torch.from_numpy(golden["ce_logits"]),
                                                         torch.from_numpy(golden["ce_soft"]))
    np.testing.assert_allclose(ce.numpy(), golden["ce_none"], rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        hv.LaplaceNLLLoss(reduction="bogus")(torch.zeros(1, 4), torch.zeros(1, 2))
    with pytest.raises(ValueError):
        hv.SoftTargetCrossEntropyLoss(reduction="bogus")(torch.zeros(1, 4), torch.zeros(1, 4))


def test_product_critics_and_masked_losses_match_golden(golden):
    """The product's critics (state loaded from the fixture) and its masked hinge / R1 / generator
    losses reproduce the reference's values, also when extra masked-out rows are appended."""
    critics = {"short": hv.ShortScaleCritic(10, 2, 16), "mid": hv.MidScaleCritic(20, 2, 24),
               "long": hv.LongScaleCritic(30, 2, 24)}
    prefix = {"short": "D_short", "mid": "D_mid", "long": "D_long"}
    for name, c in critics.items():
        c.load_state_dict({k[len(f"critic_init.{prefix[name]}."):]: torch.from_numpy(v) for k, v in golden.items()
                           if k.startswith(f"critic_init.{prefix[name]}.")})
        c.train()
        for m in c.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    real, fake = torch.from_numpy(golden["real"]), torch.from_numpy(golden["fake"]).requires_grad_(True)
    for name, c in critics.items():
        np.testing.assert_allclose(c(real).detach().numpy(), golden[f"critic_fwd1.{name}"], rtol=1e-5, atol=1e-6)
        c(real)
    pad_real = torch.cat([real, torch.randn(3, 30, 2)])
    pad_fake = torch.cat([fake, torch.randn(3, 30, 2)])
    keep = torch.tensor([True] * real.shape[0] + [False] * 3)
    d_loss, d_logs = hv.AdversarialDiscriminatorLoss(lambda_r1=10.0).forward_masked(critics, pad_real, pad_fake, keep)
    np.testing.assert_allclose(d_loss.item(), golden["d_loss"], rtol=1e-5)
    params = {f"{prefix[n]}.{k}": p for n, c in critics.items() for k, p in c.named_parameters()}
    grads = torch.autograd.grad(d_loss, list(params.values()))
    for k, g in zip(params, grads):
        np.testing.assert_allclose(g.numpy(), golden[f"d_grad.{k}"], rtol=2e-4, atol=2e-6, err_msg=k)
    g_loss, _ = hv.AdversarialGeneratorLoss(lambda_adv=0.1).forward_masked(critics, pad_fake, keep)
    (g_fake,) = torch.autograd.grad(g_loss, fake)
    np.testing.assert_allclose(g_loss.item(), golden["g_loss"], rtol=1e-5)
    np.testing.assert_allclose(g_fake.numpy(), golden["g_grad_fake"], rtol=1e-5, atol=1e-7)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "hivt_mi355.h")).read()
    declared = set(re.findall(r"\b(hivt_[a-z0-9_]+)\s*\(", header))
    declared -= {"hivt_last_error"} - {"hivt_last_error"}
    assert len(declared) >= 18
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/hivt_mi355.h but not exported"
    assert set(_lib.SIGNATURES) == declared
    assert _lib.query("hivt_version") >= 100
    assert _lib.query("hivt_csr_workspace_bytes", 1000, 10) % 256 == 0
    assert _lib.load().hivt_last_error() == b""


def test_generated_binding_covers_the_same_entry_points():
    """The generated CPython binding must expose every entry point of the header, answer the size queries
    like the ctypes binding, and raise the library's error type with the library's message."""
    fast = _lib._load_fast()
    assert fast, "hivt_gan_amd/_hivt_fast.so has not been built (make -C hivt_gan_amd/csrc)"
    lib = _lib.load()
    for name in _lib.SIGNATURES:
        assert hasattr(fast, name), f"{name} has no generated wrapper"
    assert fast.hivt_version() == lib.hivt_version()
    widths = (ctypes.c_int32 * 3)(20, 128, 128)
    assert fast.hivt_critic_workspace_bytes(100, 2, widths) == lib.hivt_critic_workspace_bytes(100, 2, widths)
    assert fast.hivt_linear_wgrad_workspace_bytes(1000, 128, 128) == lib.hivt_linear_wgrad_workspace_bytes(1000, 128, 128)
    # argument checks run before anything touches a GPU: an unknown epilogue code is refused
    args = [0, 128, 0, 128, None, None, 0, None, 0, None, None, 1e-5, 9, 1.0, 0.0, 0, None, None, 0, 5, 128, 128, 0, 128,
            None, 0, None, 0]
    with pytest.raises(_lib.HivtLibraryError, match="unknown epilogue 9"):
        fast.hivt_linear_fwd(*args)
    with pytest.raises(TypeError):
        fast.hivt_linear_fwd(*args[:-1])
    args[12] = 0
    args[19] = 0  # no rows: returns before any launch
    assert fast.hivt_linear_fwd(*args) is None


def test_generated_binding_refuses_ctypes_pointer_objects():
    """A ``c_void_p`` exposes the eight host bytes holding the address as a buffer; taken as a host table it would
    hand a kernel a host address.  The binding takes ints, None and arrays of numbers only."""
    fast = _lib._load_fast()
    if not fast:
        pytest.skip("the generated binding is not built")
    with pytest.raises(TypeError):
        fast.hivt_linear_wgrad_group(ctypes.c_void_p(64), 0, None, 0, None)
    with pytest.raises(TypeError):
        fast.hivt_linear_wgrad_group(ctypes.pointer(ctypes.c_int64(0)), 0, None, 0, None)
    assert fast.hivt_linear_wgrad_group((ctypes.c_int64 * 10)(), 0, None, 0, None) is None  # no terms: no GPU work


def test_kernels_refuse_cpu_tensors():
    from hivt_gan_amd import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.csr_build(torch.zeros(4, dtype=torch.int32), 3)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.gated_update(torch.zeros(2, 8), torch.zeros(2, 8), torch.zeros(2, 16))


def test_seed_scope_is_reproducible_and_nests():
    import torch
    from hivt_gan_amd import ops

    def draw():
        torch.manual_seed(77)
        with ops.seed_scope():
            first = [ops._fresh_seed() for _ in range(5)]
            with ops.seed_scope():
                inner = ops._fresh_seed()
            first.append(ops._fresh_seed())  # the outer counter continues after the nested scope
        return first, inner

    a, inner_a = draw()
    b, inner_b = draw()
    assert a == b and inner_a == inner_b
    assert len(set(a + [inner_a])) == 7 and all(0 <= s < 2 ** 62 for s in a)
    assert ops._scope_base is None  # outside a scope every call draws from the generator again
    torch.manual_seed(77)
    outside = ops._fresh_seed()
    torch.manual_seed(77)
    assert ops._fresh_seed() == outside


def test_shape_buckets_hold_their_batches_and_beat_one_shape():
    """``BatchSpec.shape_buckets``: every batch lands in a bucket whose shape holds it, the cuts never cost more
    padded rows than one shape for all or than equal-count runs, and the number of buckets is what was asked for."""
    import random
    rng = random.Random(7)
    specs = []
    for _ in range(32):
        n = rng.randint(1900, 3300)
        up = lambda v, m: -(-v // m) * m  # noqa: E731
        specs.append(hv.BatchSpec(up(n + 1, 64), up(3 * n, 64), up(30 * n + rng.randint(0, 900), 64), up(50 * n, 64),
                                  edge_capacity=(up(25 * n, 256), up(48 * n + rng.randint(0, 5000), 256), up(18 * n, 256),
                                                 up(10 * n, 256))))
    shapes, which = hv.BatchSpec.shape_buckets(specs, 4)
    assert len(shapes) == 4 and len(which) == 32 and set(which) == {0, 1, 2, 3}
    for sp, j in zip(specs, which):
        assert all(a <= b for a, b in zip(sp.key(), shapes[j].key()))
    cost = sum(shapes[j].padded_work() for j in which)
    one = 32 * hv.BatchSpec.union(specs).padded_work()
    order = sorted(range(32), key=lambda i: specs[i].padded_work())
    quartiles = sum(8 * hv.BatchSpec.union([specs[i] for i in order[a:a + 8]]).padded_work() for a in range(0, 32, 8))
    assert cost <= quartiles < one
    assert cost >= sum(sp.padded_work() for sp in specs)
    # fewer batches than buckets; equal shapes merge unless every rank must make the same number of recordings
    shapes, which = hv.BatchSpec.shape_buckets(specs[:2], 4)
    assert len(shapes) == 2 and sorted(which) == [0, 1]
    shapes, which = hv.BatchSpec.shape_buckets([specs[0]] * 5, 4)
    assert len(shapes) == 1 and which == [0] * 5
    shapes, which = hv.BatchSpec.shape_buckets([specs[0]] * 5, 4, merge_equal=False)
    assert len(shapes) == 4 and sorted(set(which)) == [0, 1, 2, 3]
    assert hv.BatchSpec.shape_buckets([], 4) == ([], [])
