#!/usr/bin/env python3
"""SURVEY's cold C3 (T in [-5, 15]): how the attempts are distributed over the hillslopes and over the waves --
what keeping the lanes busy could gain at most (sum over waves of the slowest lane / sum over lanes).
Run on the GPU box: python tools/cold_hist.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tiger_hlm_gpu_amd as H
from tiger_hlm_gpu_amd import synth
from tiger_hlm_gpu_amd.engine import ShardEngine
ns, days = 1_000_000, 30
sp = synth.model204_params(0, ns); y0 = synth.model204_y0(ns); cells = synth.cell_of_system(0, ns)
pr, t2m, dt, nT = synth.forcing_tables(days=days, tmin=-5.0)
e = ShardEngine(H.Model204, y0, 0.0, days*1440.0, sp=sp)
e.expand_forcings_on_device((pr, t2m), cells, dt, nT)
e.integrate(); r = e.results()
na = r["n_attempt"].astype(np.int64); st = r["stiff"]
print("flagged", (st==1).mean(), "attempts: mean", na.mean(), "max", na.max(), "quantiles", np.quantile(na,[0.5,0.9,0.99,0.999,1.0]))
print("flagged attempts quantiles", np.quantile(na[st==1],[0.1,0.5,0.9,0.99,1.0]), "ok attempts quantiles", np.quantile(na[st==0],[0.01,0.5,0.99,1.0]))
w = na[:ns//64*64].reshape(-1,64)
print("per-wave max: mean", w.max(axis=1).mean(), "max", w.max(axis=1).max(), " per-wave mean:", w.mean(axis=1).mean(), " sum of wave max / sum", w.max(axis=1).sum()*64/ na.sum())
