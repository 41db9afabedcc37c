#!/usr/bin/env python3
"""Launch shape of the headline pass: one wave per tile (time_pieces 0) against the persistent grid over (time piece,
tile) items (csrc/hlm_rk45_queue_kernel.h) for several piece counts, same box, same inputs; every result compared
bit for bit with the fixed map's.   python tools/pieces_probe.py [--ns N] [--days D] [--pieces 0 2 4 8 ...]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import tiger_hlm_gpu_amd as H  # noqa: E402
from tiger_hlm_gpu_amd import synth  # noqa: E402
from tiger_hlm_gpu_amd.engine import ShardEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ns", type=int, default=1_000_000)
ap.add_argument("--days", type=int, default=30)
ap.add_argument("--tmin", type=float, default=0.5)
ap.add_argument("--wet", action="store_true")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--pieces", type=int, nargs="*", default=[0, 2, 4, 8, 16, 32, 0])
ap.add_argument("--runs", type=int, nargs="*", default=[0], help="hlm_device_job.queue_run_pieces per piece count: 0 automatic, 1 every piece handed over, n runs of n pieces")
ap.add_argument("--groups", type=int, nargs="*", default=[0], help="hlm_device_job.queue_group_tiles per piece count")
a = ap.parse_args()
sp = synth.model204_params(0, a.ns)
y0 = synth.model204_y0(a.ns, wet_surface=a.wet)
cells = synth.cell_of_system(0, a.ns)
pr, t2m, dt, nT = synth.forcing_tables(days=a.days, tmin=a.tmin)
tf = a.days * 1440.0
ref = None
for p, rp, gt in [(p, rp, gt) for p in a.pieces for rp in (a.runs if p > 1 else [0]) for gt in (a.groups if p > 1 else [0])]:
    eng = ShardEngine(H.Model204, y0, 0.0, tf, sp=sp, math_mode=H.MATH_FAST, time_pieces=p, queue_run_pieces=rp, queue_group_tiles=gt)
    eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)
    eng.integrate()
    torch.cuda.synchronize()
    ms = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.integrate()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    got = {k: v.clone() for k, v in (("y", eng.y_final), ("s", eng.stiff), ("a", eng.n_attempt), ("c", eng.n_accept))}
    same = "reference" if ref is None else all(torch.equal(got[k], ref[k]) for k in got)
    if ref is None:
        ref = got
    att = eng.total_attempts()
    print(f"time_pieces {p:3d} run_pieces {rp:2d} group_tiles {gt:2d}: ms {np.mean(ms):8.3f} (min {min(ms):8.3f})  steps/s {att / (np.mean(ms) * 1e-3):.4e}  bit-identical to the fixed map: {same}", flush=True)
    del eng
    torch.cuda.empty_cache()
