"""Measured roofs on the box: device-to-device copy (HBM), host<->device link with page-locked memory."""
import time
import torch

def bw(fn, nbytes, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t) / 1e9

n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
print(f"D2D copy 1 GiB: {bw(lambda: b.copy_(a), 2 * n):.0f} GB/s (read + write)")
x = torch.empty(n // 4, dtype=torch.float32, device="cuda")
print(f"fill 1 GiB:     {bw(lambda: x.fill_(1.0), n):.0f} GB/s (write only)")
print(f"sum 1 GiB:      {bw(lambda: x.sum(), n):.0f} GB/s (read only)")
for mb in (64, 256, 1024):
    m = mb << 20
    h = torch.empty(m, dtype=torch.uint8).pin_memory(); d = torch.empty(m, dtype=torch.uint8, device="cuda")
    print(f"{mb:5d} MiB pinned: H2D {bw(lambda: d.copy_(h, non_blocking=True), m):.1f} GB/s   D2H {bw(lambda: h.copy_(d, non_blocking=True), m):.1f} GB/s")
