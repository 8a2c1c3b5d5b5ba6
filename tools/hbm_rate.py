"""What the box's HBM delivers to plain streaming kernels of the framework (context for the roofline
fractions): a read-only reduction, a copy and a three-stream add over buffers far larger than the
last-level cache."""
import torch

dev = "cuda:0"
n = 1 << 29  # 2 GiB of float32 per buffer
a = torch.randn(n, device=dev)
b = torch.randn(n, device=dev)
c = torch.empty_like(a)


def rate(fn, nbytes, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return nbytes * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12


print(f"sum  (read 2 GiB)          {rate(lambda: a.sum(), 4 * n):.2f} TB/s")
print(f"copy (read + write 4 GiB)  {rate(lambda: c.copy_(a), 8 * n):.2f} TB/s")
print(f"add  (2 reads + write)     {rate(lambda: torch.add(a, b, out=c), 12 * n):.2f} TB/s")
