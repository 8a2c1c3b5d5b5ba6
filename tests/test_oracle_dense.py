"""Second opinion on the part of the oracle that no reference fixture pins: the edge attention that
torch_geometric evaluates with gather / scatter primitives (``oracle._edge_attention``,
``segment_softmax``, ``scatter_sum``) is re-derived here in DENSE form -- a [targets, sources] score
matrix per head, ``-inf`` where there is no edge, ``torch.softmax`` over the sources, rows without any
edge set to zero -- with no code shared with the oracle's sparse path, on the synthetic scenes."""
import math

import pytest
import torch

import hivt_gan_amd as hv
from oracle import hivt_oracle as O


def dense_attention(q_nodes, key_e, value_e, src, dst, num_dst, num_src, heads):
    """out[i] = sum_j softmax_j(q_i . k_(j->i) / sqrt(d)) v_(j->i) over the sources j with an edge j -> i."""
    d_model = key_e.shape[1]
    d = d_model // heads
    k = torch.zeros(num_dst, num_src, heads, d, dtype=key_e.dtype)
    v = torch.zeros_like(k)
    adj = torch.zeros(num_dst, num_src, dtype=torch.bool)
    k[dst, src] = key_e.view(-1, heads, d)
    v[dst, src] = value_e.view(-1, heads, d)
    adj[dst, src] = True
    scores = torch.einsum("ihd,ijhd->ijh", q_nodes.view(num_dst, heads, d), k) / math.sqrt(d)
    scores = scores.masked_fill(~adj.unsqueeze(-1), float("-inf"))
    has_edge = adj.any(dim=1)
    prob = torch.softmax(scores[has_edge], dim=1)
    out = torch.zeros(num_dst, heads, d, dtype=key_e.dtype)
    out[has_edge] = torch.einsum("ijh,ijhd->ihd", prob, v[has_edge])
    return out.reshape(num_dst, d_model)


@pytest.mark.parametrize("heads,width", [(8, 64), (2, 16)])
def test_sparse_edge_attention_equals_the_dense_form(heads, width):
    b = hv.make_batch(3, first_index=20)
    n = int(b.num_nodes)
    gen = torch.Generator().manual_seed(5)
    # agent-agent edges of one timestep (some targets have no incoming edge) and the last-step interaction set
    empty_rows = 0
    for t, radius in ((7, 50.0), (19, None), (10, 30.0)):
        ei, _ = O.temporal_edges({k: b[k] for k in b.keys()}, t, radius)
        src, dst = ei[0], ei[1]
        assert ei.shape[1] > 0
        empty_rows += int((torch.bincount(dst, minlength=n) == 0).sum())
        q = torch.randn(n, width, generator=gen, dtype=torch.float64)
        key = torch.randn(ei.shape[1], width, generator=gen, dtype=torch.float64)
        val = torch.randn(ei.shape[1], width, generator=gen, dtype=torch.float64)
        sparse = O._edge_attention(q.index_select(0, dst), key, val, dst, n, heads, 0.0, False)
        dense = dense_attention(q, key, val, src, dst, n, n, heads)
        assert torch.allclose(sparse, dense, rtol=1e-12, atol=1e-13)
        assert float(sparse[torch.bincount(dst, minlength=n) == 0].abs().max()) == 0.0
    assert empty_rows > 0  # targets without an incoming edge are part of the check
    # bipartite lane -> actor edges
    ei, _ = O.distance_drop_edge(b["lane_actor_index"], b["lane_actor_vectors"], 30.0)
    nl = b["lane_vectors"].shape[0]
    q = torch.randn(n, width, generator=gen, dtype=torch.float64)
    key = torch.randn(ei.shape[1], width, generator=gen, dtype=torch.float64)
    val = torch.randn(ei.shape[1], width, generator=gen, dtype=torch.float64)
    sparse = O._edge_attention(q.index_select(0, ei[1]), key, val, ei[1], n, heads, 0.0, False)
    dense = dense_attention(q, key, val, ei[0], ei[1], n, nl, heads)
    assert torch.allclose(sparse, dense, rtol=1e-12, atol=1e-13)


def test_global_interactor_layer_equals_a_dense_restatement():
    """The whole layer (models/global_interactor.py:99-142) written densely: keys and values are
    [target, source] tensors of node term + edge term; the direction of an edge (row 0 = source j feeds
    row 1 = target i) and the use of norm1(x) in the update are part of what is checked."""
    torch.manual_seed(1)
    D, H = 32, 4
    b = hv.make_batch(2, first_index=60)
    n = int(b.num_nodes)
    layer = hv.GlobalInteractorLayer(embed_dim=D, num_heads=H, dropout=0.0)
    layer.apply(hv.init_weights)
    with torch.no_grad():
        for p in layer.parameters():  # biases and norms away from their trivial initial values
            p.add_(0.1 * torch.randn_like(p))
    sd = {f"m.{k}": v.detach().double() for k, v in layer.state_dict().items()}
    ei = O.subgraph(~b["padding_mask"][:, 19], b["edge_index"])
    x = torch.randn(n, D, dtype=torch.float64)
    r = torch.randn(ei.shape[1], D, dtype=torch.float64)
    want = O.global_interactor_layer(sd, "m", x, ei, r, H)

    def lin(name, t):
        return t @ sd[f"m.{name}.weight"].t() + sd[f"m.{name}.bias"]

    def ln(name, t):
        mu = t.mean(-1, keepdim=True)
        var = ((t - mu) ** 2).mean(-1, keepdim=True)
        return (t - mu) / torch.sqrt(var + 1e-5) * sd[f"m.{name}.weight"] + sd[f"m.{name}.bias"]

    xn = ln("norm1", x)
    src, dst = ei[0], ei[1]
    key_e = lin("lin_k_node", xn)[src] + lin("lin_k_edge", r)
    val_e = lin("lin_v_node", xn)[src] + lin("lin_v_edge", r)
    agg = dense_attention(lin("lin_q_node", xn), key_e, val_e, src, dst, n, n, H)
    gate = torch.sigmoid(lin("lin_ih", agg) + lin("lin_hh", xn))
    upd = agg + gate * (lin("lin_self", xn) - agg)
    y = x + lin("out_proj", upd)
    yn = ln("norm2", y)
    y = y + lin("mlp.3", torch.relu(lin("mlp.0", yn)))
    assert torch.allclose(want, y, rtol=1e-11, atol=1e-12)
