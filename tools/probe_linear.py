"""Per-block timeline of the forward linear kernel (tuning aid).

Build first:  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DHIVT_PROBE -Iinclude -shared \
              hivt_gan_amd/csrc/linear.hip hivt_gan_amd/csrc/api.hip -o tools/probe/libhivt_probe.so
Stamps per block (100 MHz clock): 0 start, 1 first tile staged, 2 reduction loop done, 3 stores issued.
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hivt_gan_amd import _lib  # noqa: E402  (argument types of hivt_linear_fwd)

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.environ.get("PROBE_LIB", os.path.join(here, "probe", "libhivt_probe.so")))
lib.hivt_linear_fwd.restype = ctypes.c_int
lib.hivt_linear_fwd.argtypes = _lib.SIGNATURES["hivt_linear_fwd"][1]
lib.hivt_probe_set.argtypes = [ctypes.c_void_p]

dev = "cuda:0"
torch.manual_seed(0)
if os.environ.get("PROBE_BF16"):  # operand-precision mask of the products (14 = the library's bf16 mode)
    lib.hivt_set_matmul_bf16.argtypes = [ctypes.c_int]
    lib.hivt_set_matmul_bf16(int(os.environ["PROBE_BF16"]))
PAD = int(os.environ.get("PROBE_PAD", "0"))  # extra floats per row of x and w (breaks power-of-two strides)
shapes = [(136_000, 128, 256), (136_000, 128, 128), (55_000, 512, 128), (2623, 128, 128), (2623, 512, 256)]
if os.environ.get("PROBE_SMALL"):
    shapes = [(2623, 128, 128), (2623, 128, 256), (2623, 256, 256), (2623, 128, 512), (2623, 512, 128)]
for (M, K, N) in shapes:
    x = torch.randn(M, K + PAD, device=dev)
    w = torch.randn(N, K + PAD, device=dev)
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    bm = 128 if N <= 64 else 64
    bn = 64 if N <= 64 else 128
    blocks = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
    probe = torch.zeros(blocks * 8, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        rc = lib.hivt_linear_fwd(x.data_ptr(), K + PAD, w.data_ptr(), K + PAD, b.data_ptr(), None, 0, None, 0, None, None, 0.0, 0,
                                 1.0, 0.0, 0, None, None, 0, M, N, K, y.data_ptr(), N, None, 0, None, stream)
        assert rc == 0

    assert lib.hivt_probe_set(None) == 0
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    assert lib.hivt_probe_set(probe.data_ptr()) == 0
    run()
    torch.cuda.synchronize()
    assert lib.hivt_probe_set(None) == 0
    p = probe.cpu().numpy().reshape(blocks, 8).astype(np.int64)
    t = (p[:, :4] - p[:, 0].min()) / 100.0  # microseconds
    stage, loop, store, life = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
    q = lambda v: "p10 %.2f med %.2f p90 %.2f" % tuple(np.percentile(v, [10, 50, 90]))  # noqa: E731
    print(f"M={M} K={K} N={N} blocks={blocks} span {t[:, 3].max():.1f} us")
    print("  stage first tile:", q(stage))
    print("  reduction loop  :", q(loop), f"({K // 32} steps)")
    print("  stores issued   :", q(store))
    print("  block lifetime  :", q(life))
    # per compute unit: blocks resident at a time and the gap between one block's end and the next start
    cu = ((p[:, 5] & 0xF) << 16) | (p[:, 4] & 0xFF00)  # XCC | SE, SH, CU bits of HW_ID
    ids, counts = np.unique(cu, return_counts=True)
    busy = []
    for c in ids[:64]:
        rows = t[cu == c]
        order = np.argsort(rows[:, 0])
        rows = rows[order]
        covered = np.zeros(int(t[:, 3].max() * 100) + 2, dtype=np.int32)
        for r in rows:
            covered[int(r[0] * 100):int(r[3] * 100) + 1] += 1
        busy.append(covered[covered > 0].mean())
    print(f"  compute units seen {len(ids)}, blocks per unit {counts.min()}..{counts.max()}, "
          f"resident blocks while busy {np.mean(busy):.2f}")
    first = np.argsort(t[:, 0])[:4]
    print("  first blocks:", [tuple(np.round(t[i], 2)) for i in first])
    sys.stdout.flush()
