#!/usr/bin/env python3
"""Development probe: hlm_rk45_integrate_forcing_streamed on the headline workload (forcing series in pinned host
memory, per stream or -- HLM_PROBE_PER_CELL=1 -- per cell), several chunk lengths, against the resident run; wall clock
and the GPU time between two events around the call.   python tools/streamed_probe.py [ns] [days]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tiger_hlm_gpu_amd as H
from tiger_hlm_gpu_amd import synth
from tiger_hlm_gpu_amd.engine import ShardEngine

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
days = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sp, y0 = synth.model204_params(0, ns), synth.model204_y0(ns)
cells = synth.cell_of_system(0, ns)
pr, t2m, dt, nT = synth.forcing_tables(days=days)
PER_CELL = os.environ.get("HLM_PROBE_PER_CELL") == "1"   # tiny tables: the uploads cost nothing
tables = [ShardEngine.pin_forcing_table(t) for t in ((pr, t2m) if PER_CELL else (pr[:, cells], t2m[:, cells]))]
tf = days * 1440.0
ref = ShardEngine(H.Model204, y0, 0.0, tf, sp=sp)
ref.expand_forcings_on_device((pr, t2m), cells, dt, nT)
for _ in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ref.integrate(); e1.record(); torch.cuda.synchronize()
print(f"resident: {e0.elapsed_time(e1):.1f} ms")
want = ref.results()
del ref
torch.cuda.empty_cache()
for pieces in [int(x) if x != "auto" else None for x in os.environ.get("HLM_PROBE_PIECES", "auto").split(",")]:
    for chunk in (1440.0,) if os.environ.get("HLM_PROBE_SHORT") else (1440.0, -1440.0, -720.0, -2880.0, 5760.0, 11520.0):
        eng = ShardEngine(H.Model204, y0, 0.0, tf, sp=sp, time_pieces=pieces)
        for rep in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            eng.integrate_forcing_streamed_abi(tables, dt, chunk, cells=cells if PER_CELL else None)
            e1.record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
        got = eng.results()
        same = all(np.array_equal(got[k], want[k]) for k in ("y_final", "stiff", "n_attempt", "n_accept"))
        print(f"streamed, time_pieces {pieces}, chunk {chunk:6.0f} min: wall {wall:7.1f} ms, events {e0.elapsed_time(e1):7.1f} ms, bit-identical to resident: {same}")
        del eng
