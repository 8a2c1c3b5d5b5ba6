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
