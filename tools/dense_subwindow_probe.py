#!/usr/bin/env python3
"""The in-HBM dense kernel (tile layout, 1-minute queries over one day) by hlm_device_job.dense_sub_windows: one wave per
tile against the persistent grid over (sub-window, tile) items; every result compared bit for bit with n = 1.
python tools/dense_subwindow_probe.py [--ns N ...] [--sub 1 2 4 8]"""
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
ap.add_argument("--ns", type=int, nargs="*", default=[1_000_000, 500_000])
ap.add_argument("--sub", type=int, nargs="*", default=[1, 2, 4, 8, 16])
ap.add_argument("--tmin", type=float, default=0.5)
a = ap.parse_args()
for ns in a.ns:
    sp, y0 = synth.model204_params(0, ns), synth.model204_y0(ns)
    cells = synth.cell_of_system(0, ns)
    pr, t2m, dt, nT = synth.forcing_tables(days=1, tmin=a.tmin)
    q = synth.minute_queries(0.0, 1440.0, 1.0)
    ref = None
    for n in a.sub:
        eng = ShardEngine(H.Model204, y0, 0.0, 1440.0, sp=sp, query=q, math_mode=H.MATH_FAST, dense_tiled=2, dense_sub_windows=n)
        eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)
        for _ in range(2):
            eng.integrate()
        torch.cuda.synchronize()
        ms = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.integrate(); e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        got = (eng.dense.clone(), eng.y_final.clone(), eng.stiff.clone(), eng.n_attempt.clone())
        same = "reference" if ref is None else all(torch.equal(x, y) for x, y in zip(got, ref))
        if ref is None:
            ref = got
        gb = 40.0 * ns * len(q) / 1e9
        print(f"ns {ns:8d} sub-windows {n:3d}: ms {np.mean(ms):7.3f} (min {min(ms):7.3f})  {gb / np.mean(ms):6.3f} TB/s = {gb / np.mean(ms) / 8.0:.4f} of the HBM spec  bit-identical to n = 1: {same}", flush=True)
        del eng, got
        torch.cuda.empty_cache()
    del ref
    torch.cuda.empty_cache()
