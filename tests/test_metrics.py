"""Validation metrics: every class against a plain-loop restatement of the reference's formula
(``metrics/*.py``; the reference classes need torchmetrics, which is not installed: parity unpinned),
accumulation over several updates, and the two-rank sum of the states."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import hivt_gan_amd as hv


def _inputs(seed, b=7, t=30, f=6):
    g = torch.Generator().manual_seed(seed)
    pred = torch.randn(b, t, 2, generator=g).cumsum(dim=1) * 3.0
    target = torch.randn(b, t, 2, generator=g).cumsum(dim=1) * 3.0
    modes = torch.randn(f, b, t, 2, generator=g).cumsum(dim=2)
    return pred, target, modes


def test_metrics_match_loop_restatements_and_accumulate():
    ade, fde, mr, jerk, sv, div = hv.ADE(), hv.FDE(), hv.MR(), hv.Jerk(), hv.SpeedViolation(speed_limit=6.0), \
        hv.EndpointDiversity()
    exp = {k: [0.0, 0] for k in ("ade", "fde", "mr", "jerk", "sv", "div")}
    for seed in (1, 2, 3):
        pred, target, modes = _inputs(seed, b=5 + seed)
        ade.update(pred, target)
        fde.update(pred, target)
        mr.update(pred, target)
        jerk.update(pred)
        sv.update(pred)
        div.update(modes)
        p, y, m = pred.double().numpy(), target.double().numpy(), modes.double().numpy()
        for i in range(p.shape[0]):
            d = [math.hypot(*(p[i, s] - y[i, s])) for s in range(p.shape[1])]
            exp["ade"][0] += sum(d) / len(d)
            exp["fde"][0] += d[-1]
            exp["mr"][0] += float(d[-1] > 2.0)
            vel = p[i, 1:] - p[i, :-1]
            acc = vel[1:] - vel[:-1]
            jk = acc[1:] - acc[:-1]
            exp["jerk"][0] += float((jk ** 2).sum(-1).mean())
            exp["sv"][0] += float((np.hypot(vel[:, 0], vel[:, 1]) > 6.0).any())
        for k in ("ade", "fde", "mr", "jerk", "sv"):
            exp[k][1] += p.shape[0]
        total, pairs = 0.0, 0
        for i in range(m.shape[0]):
            for j in range(i + 1, m.shape[0]):
                e = m[i, :, -1] - m[j, :, -1]
                total += float(np.hypot(e[:, 0], e[:, 1]).mean())
                pairs += 1
        exp["div"][0] += total / pairs
        exp["div"][1] += 1
    for metric, key in ((ade, "ade"), (fde, "fde"), (mr, "mr"), (jerk, "jerk"), (sv, "sv"), (div, "div")):
        want = exp[key][0] / exp[key][1]
        assert abs(float(metric.compute()) - want) <= 1e-5 * max(1.0, abs(want)), key
    assert 0.0 < float(mr.compute()) <= 1.0
    jerk.update(torch.zeros(3, 3, 2))           # fewer than four steps: ignored
    div.update(torch.zeros(1, 4, 5, 2))         # a single mode: ignored
    assert abs(float(jerk.compute()) - exp["jerk"][0] / exp["jerk"][1]) <= 1e-5 * exp["jerk"][0]
    ade.reset()
    assert math.isnan(float(ade.compute()))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pred, target, modes = _inputs(10 + rank, b=4 + 3 * rank)
        vm = hv.ValidationMetrics()
        vm.minADE.update(pred, target)
        vm.minMR.update(pred, target)
        vm.endpoint_diversity.update(modes)
        vm.sync()
        if rank == 0:
            torch.save({k: v for k, v in vm.compute().items()}, os.path.join(out_dir, "m.pt"))
    finally:
        dist.destroy_process_group()


def test_metric_states_add_up_across_two_ranks(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(str(tmp_path), "m.pt"))
    ade, mr, div = hv.ADE(), hv.MR(), hv.EndpointDiversity()
    for rank in (0, 1):
        pred, target, modes = _inputs(10 + rank, b=4 + 3 * rank)
        ade.update(pred, target)
        mr.update(pred, target)
        div.update(modes)
    assert torch.allclose(got["val_minADE"], ade.compute())
    assert torch.allclose(got["val_minMR"], mr.compute())
    assert torch.allclose(got["val_endpoint_diversity"], div.compute())
    assert math.isnan(float(got["val_jerk"]))  # never updated


def _metrics_worker(rank, world, port):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hivt_gan_amd.metrics import ValidationMetrics
        m = ValidationMetrics()
        if rank == 0:  # the other rank has seen no validation batch at all
            gen = torch.Generator().manual_seed(0)
            pred, target = torch.randn(5, 30, 2, generator=gen), torch.randn(5, 30, 2, generator=gen)
            m.minADE.update(pred, target)
            m.minFDE.update(pred, target)
            m.jerk.update(pred[:, :3])  # ignored: fewer than four steps
        m.sync()
        out = m.compute()
        vals = [None, None]
        dist.all_gather_object(vals, {k: float(v) for k, v in out.items()})
        assert vals[0]["val_minADE"] == vals[1]["val_minADE"] and vals[0]["val_minADE"] == vals[0]["val_minADE"]
        assert vals[0]["val_jerk"] != vals[0]["val_jerk"]  # NaN: no rank has a sample
    finally:
        dist.destroy_process_group()


def test_sync_with_a_rank_that_saw_nothing_stays_in_lock_step():
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_metrics_worker, args=(2, port), nprocs=2, join=True)
