#!/usr/bin/env python3
"""Development probe (needs a -DHLM_RADAU_COUNTERS build: HLM_HIP_LIB=...): where the lanes of the Radau pass wait for
each other on SURVEY's cold C3 -- wave-level against lane-level trip counts of the step loop, the Newton loop and the
accepted-step tail, restart and continue mode.   python tools/radau_divergence_probe.py [ns] [days]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiger_hlm_gpu_amd as H
from tiger_hlm_gpu_amd import synth
from tiger_hlm_gpu_amd.engine import ShardEngine

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
days = int(sys.argv[2]) if len(sys.argv) > 2 else 30
L = H._lib.lib()
sp, y0 = synth.model204_params(0, ns), synth.model204_y0(ns)
cells = synth.cell_of_system(0, ns)
pr, t2m, dt, nT = synth.forcing_tables(days=days, tmin=-5.0)
for mode in (True, "continue"):
    eng = ShardEngine(H.Model204, y0, 0.0, days * 1440.0, sp=sp, queue_run_pieces=1)
    eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)
    eng.integrate(radau=mode)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    L.hlm_debug_radau_counters(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.integrate(radau=mode); e1.record(); torch.cuda.synchronize()
    L.hlm_debug_radau_counters(out)
    w = list(out)
    names = ("step attempts", "Newton iterations", "accepted-step tails", "state-machine trips")
    print(f"mode {mode}: RK45 + Radau {e0.elapsed_time(e1):.1f} ms (with counters)")
    for k, n in enumerate(names):
        if w[2 * k]:
            print(f"   {n:22s}: wave trips {w[2*k]:>12d}, lane trips {w[2*k+1]:>14d}, lanes per trip {w[2*k+1] / w[2*k]:.1f} of 64")
    print(f"   Newton iterations per step attempt: lane {w[3] / w[1]:.2f}, wave {w[2] / w[0]:.2f}")
