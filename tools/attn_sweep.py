"""Launch-shape sweep of the segment-attention kernels on the three edge sets of the headline workload
(HiVT-128, 128 synthetic scenes): every shape reachable through the launch-shape hint is timed with the
kernel's own start / stop stamps, variants interleaved in one process, median of ``--reps`` launches.

Cache state mimics the step: the forward is preceded by the K|V projection that produces its input (the
rows come out of the product that precedes the kernel), the backward by a 512 MiB fill (in the step the
saved K|V are long gone from the caches when the backward reads them).

    python tools/attn_sweep.py [--reps 15] [--dense]
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import hivt_gan_amd as hv  # noqa: E402
from hivt_gan_amd import _lib, ops  # noqa: E402
from hivt_gan_amd.local_encoder import build_edge_plan  # noqa: E402

WAVES = {"slot": 1, "w1": 2, "w2": 3, "w4": 4}


def hint(waves: str, unroll: int) -> int:
    return (WAVES[waves] << 8) | (unroll << 12)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--scenes", type=int, default=128)
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--no-node-terms", action="store_true",
                    help="time the interaction set without the per-source node rows (what the in-kernel gathers cost)")
    ap.add_argument("--only", default=None, help="time only this edge set (agent-agent, lane-actor, interaction)")
    ap.add_argument("--uniform", action="store_true",
                    help="replace every edge set's row lengths by equal ones (same rows, same edges): what the spread "
                         "of the row lengths costs")
    ap.add_argument("--shuffled", action="store_true", help="the same row lengths in a random order")
    ap.add_argument("--sorted", action="store_true",
                    help="rows in descending order of length (same multiset of lengths): what the ORDER of the rows costs")
    ap.add_argument("--ranked", action="store_true",
                    help="hand the launches the CSR build's ranking of the rows by length (sets that carry one)")
    ap.add_argument("--half-grads", action="store_true",
                    help="backward writes dK | dV as bfloat16 rows (sets without node terms only)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    D, H = 128, 8
    kw = dict(num_agents=200, num_lanes=1000, all_valid=True, local_radius=150.0) if args.dense else {}
    batch = hv.make_batch(args.scenes if not args.dense else min(args.scenes, 4), **kw).to(dev)
    plan = build_edge_plan(batch, 20, 150.0 if args.dense else 50.0)
    gi = plan.interaction
    gi.src = gi.coo_src if gi.perm is None else gi.coo_src.index_select(0, gi.perm.long())
    torch.manual_seed(0)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    shapes = [("slot", 2), ("slot", 4), ("slot", 8), ("w1", 2), ("w1", 4), ("w1", 8), ("w2", 2), ("w2", 4), ("w2", 8),
              ("w4", 2), ("w4", 4), ("w4", 8)]
    for tag, es, node_terms in (("agent-agent", plan.agent_agent, False), ("lane-actor", plan.lane_actor, False),
                                ("interaction", plan.interaction, not args.no_node_terms)):
        if args.only is not None and tag != args.only:
            continue
        n, e = es.num_dst, es.num_edges
        if args.uniform or args.sorted or args.shuffled:
            deg = (es.rowptr[1:] - es.rowptr[:-1]).cpu().long()
            if args.uniform:
                base, extra = divmod(int(deg.sum()), n)
                deg = torch.full((n,), base, dtype=torch.long)
                deg[:extra] += 1
            elif args.shuffled:
                deg = deg[torch.randperm(n, generator=torch.Generator().manual_seed(1))]
            else:
                deg = torch.sort(deg, descending=True).values
            rowptr = torch.zeros(n + 1, dtype=torch.long)
            rowptr[1:] = deg.cumsum(0)
            es.rowptr = rowptr.to(torch.int32).to(dev)
        rp = es.rowptr.cpu()
        live = int((rp[1:] > rp[:-1]).sum())
        embed = torch.randn(e, D, device=dev)
        lin_k, lin_v = torch.nn.Linear(D, D).to(dev), torch.nn.Linear(D, D).to(dev)
        q = torch.randn(n, D, device=dev)
        kv_node = torch.randn(es.num_src, 2 * D, device=dev) if node_terms else None
        d_out = torch.randn(n, D, device=dev)
        out = torch.empty(n, D, device=dev)
        m = torch.empty(n, H, device=dev)
        den = torch.empty(n, H, device=dev)
        dq = torch.empty(n, D, device=dev)
        half = args.half_grads and not node_terms
        dkv = torch.empty(e, 2 * D, device=dev, dtype=torch.bfloat16 if half else torch.float32)
        src = es.src if node_terms else None
        fb = 4 * D
        gfb = 2 * D if half else 4 * D      # byte offset of dV inside a dK | dV row
        gflag = ops.DKV_BF16 if half else 0
        order_ptr = None
        if args.ranked:
            ranked = es.row_order
            if ranked is None:  # a set built without the ranking: the same order from the host, to price it
                ranked = torch.argsort(es.rowptr[1:] - es.rowptr[:-1], descending=True, stable=True).to(torch.int32)
            order_ptr = ranked.data_ptr()
        results = {"fwd": {}, "bwd": {}}

        def fwd(hint_bits, ev0, ev1, kv):
            _lib.call("hivt_seg_attn_fwd_timed", q.data_ptr(), D, kv.data_ptr(), kv.data_ptr() + fb, 2 * D,
                      kv_node.data_ptr() if node_terms else None, kv_node.data_ptr() + fb if node_terms else None,
                      2 * D if node_terms else 0, src.data_ptr() if node_terms else None, es.rowptr.data_ptr(), order_ptr,
                      n, e, D, H, 0.1, 1234, None, out.data_ptr(), m.data_ptr(), den.data_ptr(), hint_bits, stream.cuda_stream,
                      int(ev0.cuda_event), int(ev1.cuda_event))

        def bwd(hint_bits, ev0, ev1, kv):
            _lib.call("hivt_seg_attn_bwd_timed", d_out.data_ptr(), q.data_ptr(), D, kv.data_ptr(), kv.data_ptr() + fb, 2 * D,
                      kv_node.data_ptr() if node_terms else None, kv_node.data_ptr() + fb if node_terms else None,
                      2 * D if node_terms else 0, src.data_ptr() if node_terms else None, es.rowptr.data_ptr(),
                      order_ptr, out.data_ptr(), m.data_ptr(), den.data_ptr(), n, e, D, H, 0.1, 1234, None, dq.data_ptr(), D,
                      dkv.data_ptr(), dkv.data_ptr() + gfb, 2 * D, 0, hint_bits | gflag, stream.cuda_stream,
                      int(ev0.cuda_event), int(ev1.cuda_event))

        with torch.no_grad():
            kv_fixed = ops.pair_linear(embed, lin_k, lin_v)
            fwd(0, *(_events(stream)), kv_fixed)  # valid out / m / den for the backward runs
            for rep in range(args.reps + 2):
                for shape in [("auto", 0)] + shapes:
                    bits = 0 if shape[0] == "auto" else hint(*shape)
                    kv = ops.pair_linear(embed, lin_k, lin_v)  # the producer of the rows, right before the kernel
                    ev = _events(stream)
                    try:
                        fwd(bits, *ev, kv)
                    except _lib.HivtLibraryError:
                        continue
                    results["fwd"].setdefault(shape, []).append(ev)
                    flush_sum = flush.view(torch.int32).sum()  # read-only: evicts without leaving dirty lines
                    ev = _events(stream)
                    bwd(bits, *ev, kv_fixed)
                    results["bwd"].setdefault(shape, []).append(ev)
        torch.cuda.synchronize()
        deg = (rp[1:] - rp[:-1]).float()
        qs = torch.quantile(deg[deg > 0], torch.tensor([0.1, 0.5, 0.9, 0.99, 1.0])).tolist() if live else []
        print(f"== {tag}: {n} rows ({live} with edges), {e} edges, avg {e / max(n, 1):.1f} per row; "
              f"non-empty row length p10/p50/p90/p99/max {[int(v) for v in qs]}")
        for kind in ("fwd", "bwd"):
            by = bench.seg_attn_bytes(kind, n, e, D, H, node_terms, live, 2 if (half and kind == "bwd") else 4)
            line = []
            for shape, evs in results[kind].items():
                us = statistics.median(a.elapsed_time(b) * 1e3 for a, b in evs[2:])
                line.append((us, shape))
            base = dict((s, u) for u, s in line)[("auto", 0)]
            print(f"  {kind}: {by / 1e6:.1f} MB; auto {base:.1f} us = {by / base / 1e3 / 8000:.3f} of 8 TB/s")
            for us, shape in sorted(line)[:6]:
                print(f"     {shape[0]:>5s} x{shape[1]}: {us:7.1f} us  frac {by / us / 1e3 / 8000:.3f}")


def _events(stream):
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    ev1.record(stream)
    return ev0, ev1


if __name__ == "__main__":
    main()
