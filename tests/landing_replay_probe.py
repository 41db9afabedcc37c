#!/usr/bin/env python3
"""Why does the GPU not flag a hillslope the oracle flags (or the other way round)?  For each hillslope: the GPU run paused
at the forcing-sample boundary before the oracle's flag time (hlm_device_job.t_stop; the state record holds y, t, h),
then the ORACLE's arithmetic (oracle.rk45_step in the loop of rk45_kernel.cu:53-163) continued from the GPU's own state.
If that continuation agrees with the GPU, the disagreement is one of step SEQUENCES (where the last step before the
boundary happened to land), not of arithmetic.  A development probe beside tests/fuzz_parity.py (it uses the oracle: test infrastructure).  usage: tests/landing_replay_probe.py <math 0|1> idx [idx ...]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
import tiger_hlm_gpu_amd as H
from tiger_hlm_gpu_amd import synth
from tiger_hlm_gpu_amd.engine import ShardEngine

O.build()
days, start, tf = 30, 300_000, 30 * 1440.0
math = int(sys.argv[1])
idx = np.array([int(a) for a in sys.argv[2:]])
sp_all = synth.model204_params(start, 50000, saturating=True)
cells_all = synth.cell_of_system(start, 50000)
pr, t2m, dt, nT = synth.forcing_tables(days=days)
rtol, atol, h0 = 1e-6, 1e-9, 0.01


def replay(sp, cell, y, t, h, n, verbose=False):
    rc, log = 0, []
    while t < tf and n < 300000:
        if t + h > tf:
            h = tf - t
        F = np.array([pr[min(int(t / 60.0), nT[0] - 1), cell], t2m[min(int(t / 1440.0), nT[1] - 1), cell]], dtype=np.float32)
        n += 1
        k, yo, err = O.rk45_step(O.UID_204, t, y, h, rtol, atol, sp=sp, F=F)
        if err <= 1.0:
            rc = 0
            if np.abs(k[0] - k[1]).max() > 100.0:
                h = max(h * 0.5, h0 * 1e-6)
                continue
            log.append((n, t, h, err, "acc"))
            y, t = yo, t + h
            h *= min(max(0.9 * (1.0 / (err + 1e-16)) ** 0.2, 0.2), 10.0)
        else:
            rc += 1
            log.append((n, t, h, err, "REJ"))
            h *= min(max(min(0.9 * (1.0 / (err + 1e-16)) ** 0.2, 1.0), 0.2), 10.0)
            if rc > 5 or h < tf * 1e-6:
                return 1, t, n, log
    return 0, t, n, log


for i in idx:
    sp, cell = sp_all[i:i + 1], int(cells_all[i])
    y0 = synth.model204_y0(1, settled_for=sp)
    kw = dict(sp=sp, forc=synth.expand_per_stream(pr, t2m, np.array([cell])), forc_dt_h=dt, forc_nT=nT, attempt_cap=300000)
    ref = O.run(O.UID_204, y0, 0.0, tf, **kw)
    tflag = float(ref["t_reached"][0])
    tb = np.floor(tflag / 60.0) * 60.0
    eng = ShardEngine(H.Model204, y0, 0.0, tf, sp=sp, math_mode=math, attempt_cap=300000)
    eng.set_forcings_per_cell((pr, t2m), np.array([cell], dtype=np.int32), dt, nT)
    eng.integrate()
    full = eng.results()
    eng2 = ShardEngine(H.Model204, y0, 0.0, tf, sp=sp, math_mode=math, attempt_cap=300000)
    eng2.set_forcings_per_cell((pr, t2m), np.array([cell], dtype=np.int32), dt, nT)
    eng2.integrate(t_stop=tb)
    blob = eng2.checkpoint()
    f64 = blob[:7 * 8].view(np.float64)
    i32 = blob[7 * 8:].view(np.int32)
    y, t, h = f64[:5].copy(), float(f64[5]), float(f64[6])
    print(f"hillslope {i} math {math}: oracle flag {int(ref['stiff'][0])} at t = {tflag:.6f} after {int(ref['n_attempt'][0])} attempts; GPU flag "
          f"{int(full['stiff'][0])}, {int(full['n_attempt'][0])} attempts; GPU paused at boundary {tb:.0f}: t = {t:.6f}, h = {h:.6g}, attempts {i32[1]}, status {i32[4]}")
    flag, tend, n, log = replay(sp, cell, y, t, h, int(i32[1]))
    print(f"   oracle arithmetic continued from the GPU's state: flag {flag} at t = {tend:.6f}, {n} attempts;  first attempts:",
          [(e[0], round(e[1], 4), float('%.4g' % e[2]), float('%.4g' % e[3]), e[4]) for e in log[:8]])
