#!/usr/bin/env python3
"""SURVEY's cold C3 (T in [-5, 15]): which RHS forms the RK45 pass's lane-attempts run on, and how the waves mix them
(hlm_device_job.d_path_counts, the counting instance of the queue kernel).  Run on the GPU box: python tools/cold_paths_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiger_hlm_gpu_amd as H
from tiger_hlm_gpu_amd import synth
from tiger_hlm_gpu_amd.engine import ShardEngine
ns, days = 1_000_000, 30
sp, y0, cells = synth.model204_params(0, ns), synth.model204_y0(ns), synth.cell_of_system(0, ns)
for tmin in (-5.0, 0.5):
    pr, t2m, dt, nT = synth.forcing_tables(days=days, tmin=tmin)
    e = ShardEngine(H.Model204, y0, 0.0, days * 1440.0, sp=sp, queue_group_tiles=0)   # (the counting instance takes tiles as items)
    e.expand_forcings_on_device((pr, t2m), cells, dt, nT)
    e.count_paths(True)
    e.integrate()
    torch.cuda.synchronize()
    print(tmin, json.dumps(e.path_shares()), "flagged", float((e.stiff == 1).sum().item()) / ns, "attempts", e.total_attempts())
