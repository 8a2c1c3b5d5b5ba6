"""Grouped weight gradients (``hivt_linear_wgrad_group``) against one ``hivt_linear_wgrad`` call per term, bit for
bit, and the training step with its small weight gradients deferred to the end of each backward pass against
the step that issues them one by one."""
import pytest
import torch

import hivt_gan_amd as hv
from hivt_gan_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def single(dy, x, bias=True, accumulate_into=None):
    m, n = dy.shape
    k = x.shape[1]
    dw = torch.empty(n, k, device=DEV)
    db = torch.empty(n, device=DEV) if bias else None
    ws = torch.empty(max(_lib.query("hivt_linear_wgrad_workspace_bytes", m, n, k), 256), dtype=torch.uint8, device=DEV)
    _lib.call("hivt_linear_wgrad", dy.data_ptr(), dy.stride(0) if m > 1 else n, x.data_ptr(), x.stride(0) if m > 1 else k,
              m, n, k, dw.data_ptr(), db.data_ptr() if bias else None, ws.data_ptr(), ws.numel(),
              torch.cuda.current_stream().cuda_stream)
    return dw, db


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_group_equals_single_calls_bit_for_bit(precision):
    prev = ops.set_matmul_precision(precision)
    try:
        g = torch.Generator(device="cpu").manual_seed(3)

        def rnd(*shape):
            return torch.randn(*shape, generator=g).to(DEV)

        shapes = [(2615, 128, 128, True), (2615, 512, 128, True), (2615, 128, 512, True), (2615, 1, 256, True),
                  (2615, 256, 60, False), (700, 384, 128, True), (2615, 256, 30, True),  # K = 30: no vector loads
                  (40000, 128, 128, True),  # a long term: coarser chunks inside a group
                  (1, 64, 64, True)] + [(2615, 128, 128, True)] * 12  # more than sixteen terms: two launches
        terms, want = [], []
        for m, n, k, bias in shapes:
            dy, x = rnd(m, n), rnd(m, k)
            dw = torch.full((n, k), float("nan"), device=DEV)
            db = torch.full((n,), float("nan"), device=DEV) if bias else None
            terms.append((dy, x, dw, db, False, False))
            want.append(single(dy, x, bias))
        # a weight with two terms (score chain joined by the penalty chain of a critic) ...
        dy1, x1, dy2, x2 = rnd(2615, 256, ), rnd(2615, 256), rnd(2615, 256), rnd(2615, 256)
        dw_j, db_j = torch.empty(256, 256, device=DEV), torch.empty(256, device=DEV)
        terms += [(dy1, x1, dw_j, db_j, False, False), (dy2, x2, dw_j, None, False, True)]
        a, ab = single(dy1, x1)
        b, _ = single(dy2, x2, bias=False)
        # ... and a term added to what its outputs hold
        dy3, x3 = rnd(1300, 128), rnd(1300, 64)
        start_w, start_b = rnd(128, 64), rnd(128)
        dw_a, db_a = start_w.clone(), start_b.clone()
        terms.append((dy3, x3, dw_a, db_a, True, False))
        c, cb = single(dy3, x3)
        ops.linear_wgrad_group(terms)
        torch.cuda.synchronize()
        for (m, n, k, bias), (_, _, dw, db, _, _), (ref_w, ref_b) in zip(shapes, terms, want):
            if m >= 16384:  # a long term is summed in other chunks than the single call's: equal to rounding
                tol = 2e-2 if precision == "bf16" else 1e-5
                assert float((dw - ref_w).abs().max()) <= tol * float(ref_w.abs().max()), (m, n, k)
                assert float((db - ref_b).abs().max()) <= 1e-5 * float(ref_b.abs().max()), (m, n, k)
                continue
            assert torch.equal(dw, ref_w), (m, n, k)
            if bias:
                assert torch.equal(db, ref_b), (m, n, k)
        assert torch.equal(dw_j, a + b) and torch.equal(db_j, ab)
        assert torch.equal(dw_a, start_w + c) and torch.equal(db_a, start_b + cb)
    finally:
        ops.set_matmul_precision(prev)


def test_empty_terms_and_zero_rows():
    ops.linear_wgrad_group([])
    dw, db = torch.full((8, 4), 3.0, device=DEV), torch.full((8,), 3.0, device=DEV)
    ops.linear_wgrad_group([(torch.empty(0, 8, device=DEV), torch.empty(0, 4, device=DEV), dw, db, False, False)])
    assert float(dw.abs().max()) == 0.0 and float(db.abs().max()) == 0.0


def test_joining_term_must_follow_its_item():
    dy, x = torch.randn(64, 8, device=DEV), torch.randn(64, 4, device=DEV)
    with pytest.raises(_lib.HivtLibraryError):
        ops.linear_wgrad_group([(dy, x, torch.empty(8, 4, device=DEV), None, False, True)])


@pytest.mark.parametrize("use_gan", [True, False])
def test_step_with_deferred_weight_gradients_equals_the_plain_step(use_gan):
    def run(defer):
        torch.manual_seed(5)
        model = hv.HiVT(embed_dim=64, num_heads=8, dropout=0.1, use_gan=use_gan, num_temporal_layers=2,
                        num_global_layers=2).to(DEV)
        model.train()
        trainer = hv.GanTrainer(model, distributed=False)
        trainer.defer_weight_grads = defer
        torch.manual_seed(6)
        for i in range(3):
            logs = trainer.step(hv.make_batch(6, first_index=70 + 10 * i).to(DEV))
        torch.cuda.synchronize()
        return model, logs

    plain, logs_plain = run(False)
    deferred, logs_deferred = run(True)
    assert ops._deferred is None
    for (k, a), (_, b) in zip(plain.state_dict().items(), deferred.state_dict().items()):
        assert torch.equal(a, b), k
    for k in logs_plain:
        assert torch.equal(logs_plain[k], logs_deferred[k]), k


def test_grouped_row_sums_equal_the_single_layer_norm_backward_calls():
    """hivt_ln_bwd leaving its partial sums + one hivt_reduce_rows_group for several LayerNorms == one
    hivt_ln_bwd call each, bit for bit (same slices, same order); zero rows, accumulation and a split target."""
    import ctypes
    from hivt_gan_amd import _lib
    torch.manual_seed(3)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    cases, flat, keep = [], [], []
    for m, n in ((5, 128), (300, 64), (4000, 128), (70_000, 128), (1, 16)):
        x = torch.randn(m, n, device=dev)
        dy = torch.randn(m, n, device=dev)
        g, b = torch.randn(n, device=dev), torch.randn(n, device=dev)
        y, stats = torch.empty_like(x), torch.empty(m, 2, device=dev)
        _lib.call("hivt_ln_fwd", x.data_ptr(), n, g.data_ptr(), b.data_ptr(), 1e-5, m, n, y.data_ptr(), n,
                  stats.data_ptr(), stream)
        ws = torch.empty(_lib.query("hivt_ln_bwd_workspace_bytes", m, n), dtype=torch.uint8, device=dev)
        dz0, dg0, db0 = torch.empty_like(x), torch.empty(n, device=dev), torch.empty(n, device=dev)
        _lib.call("hivt_ln_bwd", dy.data_ptr(), n, x.data_ptr(), n, stats.data_ptr(), g.data_ptr(), b.data_ptr(), 0, m, n,
                  None, 0, dz0.data_ptr(), n, dg0.data_ptr(), db0.data_ptr(), ws.data_ptr(), ws.numel(), stream)
        rows = _lib.query("hivt_ln_bwd_partial_rows", m)
        part = torch.empty(_lib.query("hivt_ln_bwd_workspace_bytes", m, n) // 4, device=dev)
        assert part.numel() >= rows * 2 * n
        dz1, dg1, db1 = torch.empty_like(x), torch.full((n,), 7.0, device=dev), torch.full((n,), 7.0, device=dev)
        _lib.call("hivt_ln_bwd", dy.data_ptr(), n, x.data_ptr(), n, stats.data_ptr(), g.data_ptr(), b.data_ptr(), 0, m, n,
                  None, 0, dz1.data_ptr(), n, None, None, part.data_ptr(), part.numel() * 4, stream)
        flat += [part.data_ptr(), rows, 2 * n, 2 * n, dg1.data_ptr(), n, db1.data_ptr(), 0]
        keep.append(part)
        cases.append((dz0, dg0, db0, dz1, dg1, db1))
    # plain sums: accumulate into what the target holds, one target, no partial rows at all
    p = torch.randn(100, 40, device=dev)
    acc = torch.ones(40, device=dev)
    flat += [p.data_ptr(), 100, 40, 40, acc.data_ptr(), 1 << 30, 0, 1]
    none = torch.full((8,), 5.0, device=dev)
    flat += [0, 0, 8, 8, none.data_ptr(), 1 << 30, 0, 0]
    count = len(flat) // 8
    _lib.call("hivt_reduce_rows_group", (ctypes.c_int64 * len(flat))(*flat), count, stream)
    torch.cuda.synchronize()
    for dz0, dg0, db0, dz1, dg1, db1 in cases:
        assert torch.equal(dz0, dz1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.allclose(acc, 1.0 + p.sum(0), rtol=1e-5, atol=1e-5)
    assert torch.all(none == 0)
    with pytest.raises(_lib.HivtLibraryError):
        _lib.call("hivt_reduce_rows_group", (ctypes.c_int64 * 8)(p.data_ptr(), 4, 2, 40, acc.data_ptr(), 50, 0, 0), 1,
                  stream)  # a pitch below the column count


def test_deferred_products_keep_to_their_contract():
    """Inside ``deferred_weight_grads()`` a weight gradient is queued only while its parameters have no gradient yet
    (a populated ``.grad`` -- gradient accumulation -- takes the product on the spot and accumulates correctly), and a
    parameter that two nodes of one graph share makes the pass raise instead of summing unfilled buffers."""
    torch.manual_seed(0)
    lin = torch.nn.Linear(128, 128).to(DEV)
    x = torch.randn(300, 128, device=DEV)
    w_ref = torch.randn(300, 128, device=DEV)
    (torch.nn.functional.linear(x, lin.weight, lin.bias) * w_ref).sum().backward()
    want_w, want_b = lin.weight.grad.clone(), lin.bias.grad.clone()
    # (1) deferred, gradients start empty: filled at the end of the pass
    lin.weight.grad = lin.bias.grad = None
    with ops.deferred_weight_grads():
        (ops.fused_linear(x, lin.weight, lin.bias) * w_ref).sum().backward()
    assert float((lin.weight.grad - want_w).abs().max()) <= 1e-4 * float(want_w.abs().max())
    assert float((lin.bias.grad - want_b).abs().max()) <= 1e-4 * float(want_b.abs().max())
    # (2) gradients already populated: the second pass adds to them
    with ops.deferred_weight_grads():
        (ops.fused_linear(x, lin.weight, lin.bias) * w_ref).sum().backward()
    assert float((lin.weight.grad - 2 * want_w).abs().max()) <= 2e-4 * float(want_w.abs().max())
    assert float((lin.bias.grad - 2 * want_b).abs().max()) <= 2e-4 * float(want_b.abs().max())
    # (3) one parameter, two nodes of one graph: refused
    lin.weight.grad = lin.bias.grad = None
    x2 = torch.randn(300, 128, device=DEV)
    with pytest.raises(RuntimeError, match="not adopted"):
        with ops.deferred_weight_grads():
            (ops.fused_linear(x, lin.weight, lin.bias) + ops.fused_linear(x2, lin.weight, lin.bias)).sum().backward()
    # ... and fine outside the scope
    lin.weight.grad = lin.bias.grad = None
    (ops.fused_linear(x, lin.weight, lin.bias) + ops.fused_linear(x2, lin.weight, lin.bias)).sum().backward()
    ref = (x.sum(0) + x2.sum(0)).unsqueeze(0).expand(128, 128)
    assert float((lin.weight.grad - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
