#!/usr/bin/env python3
"""Development probe: fixed-step runs of the queue kernel against the one-wave-per-tile launch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiger_hlm_gpu_amd as H
from tiger_hlm_gpu_amd import synth
from tiger_hlm_gpu_amd.engine import ShardEngine

def run(p, tf, fs, mm):
    e = ShardEngine(H.DummyModel, synth.dummy_y0(0, 777), 0.0, tf, time_pieces=p, fixed_step=fs, math_mode=mm)
    e.integrate()
    return e.results()
for mm in (H.MATH_FAST, H.MATH_STRICT):
    for tf, fs in ((50.0, 0.05), (50.0, 0.0625), (8.0, 0.05)):
        a = run(0, tf, fs, mm)
        for p in (2, 3, 4, 5, 7, 16):
            b = run(p, tf, fs, mm)
            d = np.abs(a["y_final"] - b["y_final"])
            print(f"math={mm} tf={tf} h={fs} pieces={p}: lanes differing {int((d.max(axis=1) > 0).sum())}, max |d| {d.max():.3g}, "
                  f"attempts equal {np.array_equal(a['n_attempt'], b['n_attempt'])} ({a['n_attempt'][0]} vs {b['n_attempt'][0]}), accepts equal {np.array_equal(a['n_accept'], b['n_accept'])}")
