"""The two summed branch layers of an edge embedding as one product over weights laid side by side by
``hivt_stack_pairs`` (one launch per forward pass for all embeddings) against the concatenation made on the spot."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stack_pairs_equals_cat_and_add():
    from hivt_gan_amd import ops
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    pairs = []
    for n, ka, kb, bias in ((128, 128, 128, True), (64, 64, 64, True), (7, 5, 3, False), (130, 1, 257, True)):
        pairs.append((torch.randn(n, ka, device=dev), torch.randn(n, device=dev) if bias else None,
                      torch.randn(n, kb, device=dev), torch.randn(n, device=dev) if bias else None))
    for (w_a, b_a, w_b, b_b), (w, b) in zip(pairs, ops.stack_pairs(pairs)):
        assert torch.equal(w, torch.cat([w_a, w_b], dim=1))
        assert (b is None) == (b_a is None) and (b is None or torch.equal(b, b_a + b_b))


def test_model_with_stacked_pairs_equals_model_without():
    import hivt_gan_amd as hv
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    cfg = dict(embed_dim=64, num_heads=8, dropout=0.0, num_temporal_layers=1, num_global_layers=2, use_gan=False,
               rotate=True)
    model = hv.HiVT(**cfg).to(dev)
    model.train()

    def run(stack):
        batch = hv.make_batch(3, first_index=40).to(dev)
        for p in model.parameters():
            p.grad = None
        if not stack:
            model._stack_embedding_pairs = lambda: []
        try:
            y_hat, pi = model(batch)
        finally:
            model.__dict__.pop("_stack_embedding_pairs", None)
        (y_hat.square().mean() + pi.square().mean()).backward()
        return y_hat.detach(), pi.detach(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    y0, pi0, g0 = run(False)
    y1, pi1, g1 = run(True)
    assert torch.equal(y0, y1) and torch.equal(pi0, pi1) and g0.keys() == g1.keys()
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    from hivt_gan_amd.embedding import MultipleInputEmbedding
    assert not any("_stacked_pair" in m.__dict__ for m in model.modules() if isinstance(m, MultipleInputEmbedding))
