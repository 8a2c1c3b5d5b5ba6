"""The edge-attention block as one autograd node against the same block built from separate nodes (query projection,
key | value projection + attention, gated tail): forward bit for bit, gradients to rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _modules(d, node_terms, dev):
    torch.manual_seed(11)
    names = ["q", "k", "v", "ih", "hh", "self", "out"] + (["k_node", "v_node"] if node_terms else [])
    return {n: torch.nn.Linear(d, d).to(dev) for n in names}


def _csr(n, e, dev, gen):
    dst = torch.sort(torch.randint(0, n, (e,), generator=gen)).values
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.bincount(dst, minlength=n).cumsum(0)
    return rowptr.to(torch.int32).to(dev)


@pytest.mark.parametrize("node_terms", [False, True])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_attention_block_equals_its_separate_nodes(node_terms, precision):
    from hivt_gan_amd import ops
    dev = torch.device("cuda", 0)
    d, heads, n, e = 128, 8, 700, 9000
    gen = torch.Generator().manual_seed(3)
    lin = _modules(d, node_terms, dev)
    rowptr = _csr(n, e, dev, gen)
    src = torch.randint(0, n, (e,), generator=gen).to(torch.int32).to(dev)
    order = torch.sort(src.long(), stable=True)
    perm_by_src = order.indices.to(torch.int32)
    rowptr_by_src = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr_by_src[1:] = torch.bincount(src.long(), minlength=n).cumsum(0)
    rowptr_by_src = rowptr_by_src.to(torch.int32)
    center0 = torch.randn(n, d, generator=gen).to(dev)
    res0 = torch.randn(n, d, generator=gen).to(dev)
    edge0 = torch.randn(e, d, generator=gen).to(dev)
    weight = torch.randn(n, d, generator=gen).to(dev)
    prev = ops.set_matmul_precision(precision)
    try:
        def run(fused):
            for m in lin.values():
                for p in m.parameters():
                    p.grad = None
            c, r, x = (t.clone().requires_grad_(True) for t in (center0, res0, edge0))
            if fused:
                y = ops.attention_block(c, r, x, rowptr, lin["q"], lin["k"], lin["v"], lin["ih"], lin["hh"], lin["self"],
                                        lin["out"], heads, 0.0, 0.0, True, padded=False,
                                        lin_k_node=lin.get("k_node"), lin_v_node=lin.get("v_node"),
                                        src=src if node_terms else None,
                                        rowptr_by_src=rowptr_by_src if node_terms else None,
                                        perm_by_src=perm_by_src if node_terms else None)
            else:
                q = ops.fused_linear(c, lin["q"].weight, lin["q"].bias)
                if node_terms:
                    kv_node = ops.pair_linear(c, lin["k_node"], lin["v_node"])
                    kv = ops.pair_linear(x, lin["k"], lin["v"], table=kv_node, index=src, rowptr_by_index=rowptr_by_src,
                                         perm_by_index=perm_by_src)
                    agg = ops.seg_attn(q, kv, rowptr, heads, 0.0, True)
                else:
                    agg = ops.edge_attention(q, x, lin["k"], lin["v"], rowptr, heads, 0.0, True)
                y = ops.gated_update_tail(agg, c, r, lin["ih"], lin["hh"], lin["self"], lin["out"], 0.0, True)
            (y * weight).sum().backward()
            grads = {"center": c.grad, "res": r.grad, "edge": x.grad}
            for name, m in lin.items():
                grads[name + ".w"], grads[name + ".b"] = m.weight.grad.clone(), m.bias.grad.clone()
            return y.detach(), grads

        y_a, g_a = run(False)
        y_b, g_b = run(True)
    finally:
        ops.set_matmul_precision(prev)
    assert torch.equal(y_a, y_b)
    # the centre rows' gradient is one reduction over all projections instead of a sum of rounded products
    # (bf16 mode: the block keeps dK | dV as bfloat16 rows, which the separate nodes of the interaction layout do not)
    tol = 2e-6 if precision == "fp32" else (1e-2 if node_terms else 2e-3)
    for k in g_a:
        if k in ("k.b", "k_node.b") and precision == "bf16":
            # exactly zero in exact arithmetic (the softmax does not change when every key of a row moves by the same
            # vector): what the two runs hold is rounding residue, compared with the weight gradient's size instead
            assert float(g_b[k].norm()) <= 1e-2 * float(g_b[k[:-1] + "w"].norm())
            continue
        err = float((g_a[k] - g_b[k]).norm() / g_a[k].norm().clamp_min(1e-30))
        assert err <= (tol if k == "center" else 1e-6 if precision == "fp32" else tol), (k, err)


def test_attention_block_dropout_and_padding_run_and_stay_finite():
    from hivt_gan_amd import ops
    dev = torch.device("cuda", 0)
    d, heads, n, e = 128, 8, 300, 4000
    gen = torch.Generator().manual_seed(5)
    lin = _modules(d, False, dev)
    rowptr = _csr(n, e - 500, dev, gen)  # 500 rows of padding behind the live edges
    c = torch.randn(n, d, generator=gen).to(dev).requires_grad_(True)
    r = torch.randn(n, d, generator=gen).to(dev).requires_grad_(True)
    x = torch.randn(e, d, generator=gen).to(dev).requires_grad_(True)
    y = ops.attention_block(c, r, x, rowptr, lin["q"], lin["k"], lin["v"], lin["ih"], lin["hh"], lin["self"], lin["out"],
                            heads, 0.1, 0.1, True, padded=True)
    y.sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(c.grad).all() and torch.isfinite(x.grad).all()
    assert float(x.grad[e - 500:].abs().max()) == 0.0  # padded edge rows receive exact zeros


def test_segment_sum_of_bfloat16_rows_equals_the_sum_of_the_widened_rows():
    from hivt_gan_amd import _lib
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(7)
    n, e, width = 300, 5000, 256
    rows = torch.randn(e, width, generator=gen).to(dev).to(torch.bfloat16)
    src = torch.randint(0, n, (e,), generator=gen)
    order = torch.sort(src, stable=True)
    perm = order.indices.to(torch.int32).to(dev)
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.bincount(src, minlength=n).cumsum(0)
    rowptr = rowptr.to(torch.int32).to(dev)
    wide = rows.float()
    out_h = torch.empty(n, width, device=dev)
    out_f = torch.empty(n, width, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("hivt_segment_sum_rows_half", rows.data_ptr(), width, width, rowptr.data_ptr(), perm.data_ptr(), n,
              out_h.data_ptr(), width, st)
    _lib.call("hivt_segment_sum_rows", wide.data_ptr(), width, width, rowptr.data_ptr(), perm.data_ptr(), n,
              out_f.data_ptr(), width, st)
    assert torch.equal(out_h, out_f)
    ref = torch.zeros(n, width, dtype=torch.float64, device=dev).index_add_(0, src.to(dev), wide.double())
    assert float((out_f.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_shared_edge_rows_sum_their_gradient_inside_the_blocks(precision):
    """Three blocks read the same edge rows: with a sink the rows' gradient is the sum that autograd would form (the
    same products, the running sum added in the store epilogue), and rows that also have another reader get that
    reader's gradient added by the sharing node."""
    from hivt_gan_amd import ops
    dev = torch.device("cuda", 0)
    d, heads, n, e = 128, 8, 500, 6000
    gen = torch.Generator().manual_seed(13)
    blocks = [_modules(d, False, dev) for _ in range(3)]
    rowptr = _csr(n, e, dev, gen)
    center0 = torch.randn(n, d, generator=gen).to(dev)
    edge0 = torch.randn(e, d, generator=gen).to(dev)
    weight = torch.randn(n, d, generator=gen).to(dev)
    side = torch.randn(e, d, generator=gen).to(dev)
    prev = ops.set_matmul_precision(precision)
    try:
        def run(shared):
            for lin in blocks:
                for m in lin.values():
                    for p in m.parameters():
                        p.grad = None
            x = edge0.clone().requires_grad_(True)
            c = center0.clone().requires_grad_(True)
            rows, sink = ops.share_edge_rows(x) if shared else (x, None)
            h = c
            for lin in blocks:
                h = ops.attention_block(h, h, rows, rowptr, lin["q"], lin["k"], lin["v"], lin["ih"], lin["hh"],
                                        lin["self"], lin["out"], heads, 0.0, 0.0, True, edge_grad_sink=sink)
            loss = (h * weight).sum() + (rows * side).sum()
            # a pass that stops short of the shared rows (the blocks still form their sum) must leave nothing that
            # the next pass over the same graph would add
            torch.autograd.grad(loss, [c], retain_graph=True)
            loss.backward()
            assert sink is None or sink.total is None  # taken by the sharing node
            return x.grad, c.grad, blocks[0]["k"].weight.grad.clone()

        plain, shared = run(False), run(True)
    finally:
        ops.set_matmul_precision(prev)
    for a, b in zip(plain, shared):
        assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max())
