#!/usr/bin/env python3
"""Development probe: the headline pass and its wet-surface variant (no hillslope ever qualifies for the no-rain path),
kernel time of the default launch.   python tools/wet_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiger_hlm_gpu_amd as H
from tiger_hlm_gpu_amd import synth
from tiger_hlm_gpu_amd.engine import ShardEngine

ns, days = 1_000_000, 30
sp = synth.model204_params(0, ns)
cells = synth.cell_of_system(0, ns)
pr, t2m, dt, nT = synth.forcing_tables(days=days)
for wet in (False, True):
    eng = ShardEngine(H.Model204, synth.model204_y0(ns, wet_surface=wet), 0.0, days * 1440.0, sp=sp)
    eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.integrate(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"{'wet surface' if wet else 'dry (headline)'}: {best:.1f} ms")
    del eng
