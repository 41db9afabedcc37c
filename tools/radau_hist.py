#!/usr/bin/env python3
"""SURVEY's cold C3 (T in [-5, 15]): how the Radau-IIA pass's attempts are distributed over the flagged
hillslopes and over the waves of the compacted list -- what lane refill could gain there at most
(sum over waves of the slowest lane x 64 / sum over lanes).  Run on the GPU box: python tools/radau_hist.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tiger_hlm_gpu_amd as H
from tiger_hlm_gpu_amd import synth
from tiger_hlm_gpu_amd.engine import ShardEngine
ns, days = 1_000_000, 30
sp = synth.model204_params(0, ns); y0 = synth.model204_y0(ns); cells = synth.cell_of_system(0, ns)
pr, t2m, dt, nT = synth.forcing_tables(days=days, tmin=-5.0)
def run(radau):
    e = ShardEngine(H.Model204, y0, 0.0, days * 1440.0, sp=sp)
    e.expand_forcings_on_device((pr, t2m), cells, dt, nT)
    e.integrate(radau=radau)
    return e.results()
a, b = run(False), run(True)
fl = np.flatnonzero(a["stiff"] == 1)
ra = (b["n_attempt"].astype(np.int64) - a["n_attempt"].astype(np.int64))[fl]   # the pass adds its attempts
print("flagged", len(fl), "radau attempts per flagged hillslope: mean %.1f" % ra.mean(), "quantiles 1/10/50/90/99/100 %:",
      np.quantile(ra, [0.01, 0.1, 0.5, 0.9, 0.99, 1.0]))
w = ra[: len(ra) // 64 * 64].reshape(-1, 64)
print("waves", len(w), "per-wave max: mean %.1f" % w.max(axis=1).mean(), " per-wave mean: %.1f" % w.mean(axis=1).mean(),
      " sum of wave max x 64 / sum = %.3f" % (w.max(axis=1).sum() * 64 / w.sum()))
srt = np.sort(ra)[: len(ra) // 64 * 64].reshape(-1, 64)
print("if sorted by cost: sum of wave max x 64 / sum = %.3f" % (srt.max(axis=1).sum() * 64 / srt.sum()))
