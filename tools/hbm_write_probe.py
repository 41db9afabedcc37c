#!/usr/bin/env python3
"""What can a pure write stream reach on this GPU?  Fills a 57.6 GB tensor (the size of the dense
benchmark's output) with torch's fill kernel and with a copy; prints GB/s.  Context for the dense
kernel's roofline fraction: its traffic is write-only."""
import torch
n = 57_640_000_000 // 8
x = torch.empty(n, dtype=torch.float64, device="cuda:0")
def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
ms = timed(lambda: x.fill_(1.5))
print("fill   57.64 GB: %.2f ms = %.0f GB/s written" % (ms, n * 8 / ms / 1e6))
ms = timed(lambda: x.zero_())
print("zero   57.64 GB: %.2f ms = %.0f GB/s written" % (ms, n * 8 / ms / 1e6))
y = torch.empty(n // 2, dtype=torch.float64, device="cuda:0")
ms = timed(lambda: y.copy_(x[: n // 2]))
print("copy   28.82 GB: %.2f ms = %.0f GB/s read + written" % (ms, n * 8 / ms / 1e6))
ms = timed(lambda: x[: n // 2].sum())
print("reduce 28.82 GB: %.2f ms = %.0f GB/s read" % (ms, n * 4 / ms / 1e6))
