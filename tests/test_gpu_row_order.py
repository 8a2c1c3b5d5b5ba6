"""The ranking of CSR rows by length (``hivt_csr_build``'s ``row_order``) and the attention launches that start
their rows in that order: the ranking is a permutation with the long rows first, and it is a schedule only --
outputs and gradients are bit for bit those of the launches without it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _keys(n, e, gen, skew):
    """Target ids with row lengths spread widely (a few rows hold most of the edges) or evenly."""
    if not skew:
        return torch.randint(0, n, (e,), generator=gen).to(torch.int32)
    weight = torch.rand(n, generator=gen) ** 6 + 1e-4
    weight[::7] = 0.0  # empty rows
    return torch.multinomial(weight, e, replacement=True, generator=gen).to(torch.int32)


@pytest.mark.parametrize("n,e,skew", [(1, 5, False), (37, 0, False), (700, 9000, True), (5000, 40000, True),
                                      (70000, 300000, False)])
def test_rows_are_ranked_longest_first(n, e, skew):
    from hivt_gan_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(n + e)
    key = _keys(n, e, gen, skew).to(dev)
    plain = ops.csr_build(key, n)
    rowptr, perm, ranked = ops.csr_build(key, n, order=True)
    assert torch.equal(rowptr, plain[0]) and torch.equal(perm, plain[1])  # the grouping itself is unchanged
    assert ranked.dtype == torch.int32 and ranked.shape == (n,)
    assert torch.equal(torch.sort(ranked.long()).values, torch.arange(n, device=dev))  # a permutation
    length = (rowptr[1:] - rowptr[:-1]).long()
    ranked_class = torch.clamp(length[ranked.long()] // 2, max=511)  # classes of two lengths, the top one open
    assert bool((ranked_class[1:] <= ranked_class[:-1]).all())


def test_ranking_with_a_live_prefix_counts_only_live_edges():
    from hivt_gan_amd import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(5)
    n, e, live = 300, 4000, 2500
    key = _keys(n, e, gen, True).to(dev)
    count = torch.tensor([live], dtype=torch.int32, device=dev)
    rowptr, _, ranked = ops.csr_build(key, n, count, order=True)
    ref_rowptr, _, ref_ranked = ops.csr_build(key[:live].contiguous(), n, order=True)
    assert torch.equal(rowptr, ref_rowptr)
    length = (rowptr[1:] - rowptr[:-1]).long()
    assert torch.equal(torch.sort(ranked.long()).values, torch.arange(n, device=dev))
    cls = torch.clamp(length[ranked.long()] // 2, max=511)
    assert bool((cls[1:] <= cls[:-1]).all())
    assert torch.equal(torch.sort(length[ranked.long()] // 2).values, torch.sort(length[ref_ranked.long()] // 2).values)


@pytest.mark.parametrize("n,e", [(700, 9000), (3000, 120000)])
@pytest.mark.parametrize("half", [False, True])
def test_attention_in_ranked_order_is_bit_identical(n, e, half):
    from hivt_gan_amd import ops
    dev = torch.device("cuda", 0)
    d, heads = 128, 8
    gen = torch.Generator().manual_seed(e)
    key = torch.sort(_keys(n, e, gen, True)).values.to(dev)
    rowptr, _, ranked = ops.csr_build(key, n, order=True)
    lin_k, lin_v = torch.nn.Linear(d, d).to(dev), torch.nn.Linear(d, d).to(dev)
    q0 = torch.randn(n, d, generator=gen).to(dev)
    x0 = torch.randn(e, d, generator=gen).to(dev)
    weight = torch.randn(n, d, generator=gen).to(dev)
    results = []
    prev = ops.set_matmul_precision("bf16" if half else "fp32")
    try:
        for order in (None, ranked):
            q, x = q0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
            for p in (*lin_k.parameters(), *lin_v.parameters()):
                p.grad = None
            torch.manual_seed(77)  # the dropout seed of the call is drawn from the host generator
            out = ops.edge_attention(q, x, lin_k, lin_v, rowptr, heads, 0.1, True, row_order=order)
            (out * weight).sum().backward()
            results.append([out.detach(), q.grad, x.grad, lin_k.weight.grad.clone(), lin_v.bias.grad.clone()])
    finally:
        ops.set_matmul_precision(prev)
    for a, b in zip(*results):
        assert torch.equal(a, b)


def test_row_order_of_the_wrong_shape_is_rejected():
    from hivt_gan_amd import ops
    dev = torch.device("cuda", 0)
    rowptr = torch.tensor([0, 2, 4], dtype=torch.int32, device=dev)
    q = torch.randn(2, 16, device=dev)
    kv = torch.randn(4, 32, device=dev)
    with pytest.raises(ValueError):
        ops.seg_attn(q, kv, rowptr, 2, row_order=torch.zeros(3, dtype=torch.int32, device=dev))
    with pytest.raises(ValueError):
        ops.seg_attn(q, kv, rowptr, 2, row_order=torch.zeros(2, dtype=torch.int64, device=dev))
