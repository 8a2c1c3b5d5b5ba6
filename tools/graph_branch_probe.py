"""Do independent branches of a recorded HIP graph run side by side on this stack?

Two independent chains of small-grid products (2 615 rows, 41 x 1 workgroups each, about 10 us a launch)
are recorded once on one stream and once forked over two streams; the replay times tell whether the
runtime overlaps the branches.  The same two chains launched eagerly on two streams are the reference.

    python tools/graph_branch_probe.py [--links 60]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hivt_gan_amd import ops  # noqa: E402
from hivt_gan_amd.graph import node_census  # noqa: E402

DEV = torch.device("cuda", 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--links", type=int, default=60)
    ap.add_argument("--rows", type=int, default=2615)
    a = ap.parse_args()
    torch.manual_seed(0)
    w = [torch.randn(128, 128, device=DEV) * 0.05 for _ in range(2)]
    b = [torch.zeros(128, device=DEV) for _ in range(2)]
    x = [torch.randn(a.rows, 128, device=DEV) for _ in range(2)]
    side = torch.cuda.Stream(device=DEV)

    def chain(i):
        y = x[i]
        for _ in range(a.links):
            y = ops.fused_linear(y, w[i], b[i])
        return y

    def serial():
        return chain(0), chain(1)

    def forked():
        main_stream = torch.cuda.current_stream(DEV)
        side.wait_stream(main_stream)
        with torch.cuda.stream(side):
            y1 = chain(1)
        y0 = chain(0)
        main_stream.wait_stream(side)
        return y0, y1

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def record(fn):
        s = torch.cuda.Stream(device=DEV)
        s.wait_stream(torch.cuda.current_stream(DEV))
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream(DEV).wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g, stream=s):
            out = fn()
        g.instantiate()
        return g, out

    # how fast does a replay hand nodes to the GPU?  chains of one-workgroup kernels (about 2 us each)
    tiny = [torch.zeros(64, device=DEV) for _ in range(4)]
    sides = [torch.cuda.Stream(device=DEV) for _ in range(3)]

    def tiny_chains(streams, interleave):
        main_stream = torch.cuda.current_stream(DEV)
        lanes = [main_stream] + sides[:streams - 1]
        for s_ in lanes[1:]:
            s_.wait_stream(main_stream)
        n = a.links * 4 // streams
        if interleave:
            for _ in range(n):
                for t, s_ in zip(tiny, lanes):
                    with torch.cuda.stream(s_):
                        t.add_(1.0)
        else:
            for t, s_ in zip(tiny, lanes):
                with torch.cuda.stream(s_):
                    for _ in range(n):
                        t.add_(1.0)
        for s_ in lanes[1:]:
            main_stream.wait_stream(s_)

    with torch.no_grad():
        for streams in (1, 2, 4):
            for interleave in ((False,) if streams == 1 else (False, True)):
                g, _ = record(lambda: tiny_chains(streams, interleave))
                ms = timed(g.replay)
                print(f"tiny kernels, {a.links * 4} nodes over {streams} stream(s), "
                      f"{'interleaved' if interleave else 'one chain after the other'}: {ms:7.3f} ms = "
                      f"{ms * 1e3 / (a.links * 4):5.2f} us per node", flush=True)

    def forked_interleaved():
        main_stream = torch.cuda.current_stream(DEV)
        side.wait_stream(main_stream)
        y0, y1 = x[0], x[1]
        for _ in range(a.links):
            with torch.cuda.stream(side):
                y1 = ops.fused_linear(y1, w[1], b[1])
            y0 = ops.fused_linear(y0, w[0], b[0])
        main_stream.wait_stream(side)
        return y0, y1

    with torch.no_grad():
        want = [t.clone() for t in serial()]
        print(f"eager, one stream : {timed(serial):7.3f} ms", flush=True)
        print(f"eager, two streams: {timed(forked):7.3f} ms", flush=True)
        for name, fn in (("one stream", serial), ("two streams", forked), ("two, interleaved", forked_interleaved)):
            s = torch.cuda.Stream(device=DEV)
            s.wait_stream(torch.cuda.current_stream(DEV))
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream(DEV).wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(g, stream=s):
                out = fn()
            census = node_census(g)
            g.instantiate()
            ms = timed(g.replay)
            ok = all(torch.equal(p, q) for p, q in zip(out, want))
            print(f"replay, {name:16s}: {ms:7.3f} ms  nodes {census}  results {'ok' if ok else 'DIFFER'}", flush=True)


if __name__ == "__main__":
    main()
