#!/usr/bin/env python3
"""Development probe: does host-to-device copy traffic by itself slow the headline kernel down?  The resident run
(all forcings in HBM) timed alone and with an unrelated stream of 100 MB pinned H2D copies running beside it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiger_hlm_gpu_amd as H
from tiger_hlm_gpu_amd import synth
from tiger_hlm_gpu_amd.engine import ShardEngine

ns, days = 1_000_000, 30
sp, y0 = synth.model204_params(0, ns), synth.model204_y0(ns)
cells = synth.cell_of_system(0, ns)
pr, t2m, dt, nT = synth.forcing_tables(days=days)
eng = ShardEngine(H.Model204, y0, 0.0, days * 1440.0, sp=sp)
eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)
src = torch.empty(25_000_000, dtype=torch.float32).pin_memory()
dst = torch.empty(25_000_000, dtype=torch.float32, device="cuda")
copy = torch.cuda.Stream()


def timed(ncopies):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.integrate()
    e1.record()
    with torch.cuda.stream(copy):
        for _ in range(ncopies):
            dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


timed(0)
for n in (0, 30, 60, 0, 30, 60):
    print(f"{n:3d} copies of 100 MB beside the kernel ({n * 0.1:.1f} GB): kernel {timed(n):.1f} ms")

# and the rate of those copies by themselves
import time
for n in (10, 30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(copy):
        for _ in range(n):
            dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt_ = time.perf_counter() - t0
    print(f"{n} H2D copies of 100 MB from pinned memory: {n * 0.1 / dt_:.1f} GB/s")
