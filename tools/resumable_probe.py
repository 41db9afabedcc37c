#!/usr/bin/env python3
"""Development probe: the RESUMABLE instantiation of the RK45 kernel against the plain one on the headline workload,
one launch each, all forcings resident (one wave per tile in both): what carrying the pause / resume state costs."""
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
tf = days * 1440.0
eng = ShardEngine(H.Model204, y0, 0.0, tf, sp=sp, time_pieces=0)
eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)


def timed(**kw):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.integrate(**kw)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


print(f"plain kernel, one launch:              {timed():.1f} ms")
print(f"RESUMABLE kernel, one launch to tf:    {timed(t_stop=tf + 1.0):.1f} ms")
half = timed(t_stop=tf / 2)
print(f"RESUMABLE kernel, first half only:     {half:.1f} ms")
