"""Differential fuzzing of the HIP path against the CPU oracle (development tool, run on the GPU
box: python tests/fuzz_parity.py [n_seeds] [ns] [days]; lives under tests/ because it uses the oracle).  Each seed draws a benchmark-like but wider
workload -- static-store capacities that do and do not saturate (surface store goes wet), wet / dry /
mixed starts, rain frequency and intensity, infiltration capacities around the rain rates, per-cell
and per-stream layouts -- and checks flags, step counts and values at the solver tolerance in both
math modes.  Because such workloads are full of min / max kinks, "tolerance" is per hillslope the solver
tolerance plus 5x the separation of three yardstick runs of the oracle itself (FMA-contracted build,
rtol * (1 +- 1e-12)); a few hillslopes per thousand sit on a discrete event that none of the three
happens to move and are allowed (<= 0.5 %).  Exit code 1 and a report line per failing seed.
Typical result (24 seeds x 1024 hillslopes x 3 days, both modes): 0 flag mismatches in 49 152
hillslope runs, step counts within 6, <= 4 hillslopes per run beyond the bound."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiger_hlm_gpu_amd as H  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tiger_hlm_gpu_amd import synth  # noqa: E402
from tiger_hlm_gpu_amd.engine import ShardEngine  # noqa: E402


RADAU = False  # argv[4] == "radau": cold cells (negative ET stub flags hillslopes) + the Radau-IIA pass on both sides


def one(seed, ns, days):
    rng = np.random.default_rng(seed)
    sp = synth.model204_params(0, ns, seed=seed)
    sp["Hu"] = 10.0 ** rng.uniform(np.log10(3.05), 2.4, ns) * rng.choice([1.0, 1.0, 0.02], ns)  # some saturate quickly
    sp["infil"] = 10.0 ** rng.uniform(-5.5, -2.5, ns)
    sp["perco"] = 10.0 ** rng.uniform(-6, -3.5, ns)
    y0 = synth.model204_y0(ns)
    y0[:, 2] = np.where(rng.random(ns) < 0.3, rng.uniform(0, 0.02, ns), 0.0)
    if seed % 3 == 1:
        # whole waves that stay dry while their static stores overflow: every surface starts dry and the
        # infiltration capacity swallows any overflow (exercises the fall-backs of the no-rain path)
        y0[:, 2] = 0.0
        sp["infil"] = 10.0 ** rng.uniform(-1, 1, ns)
        sp["Hu"] = 10.0 ** rng.uniform(np.log10(3.05), 2.4, ns) * rng.choice([1.0, 0.9, 0.5], ns)
    y0[:, 0] = rng.uniform(0, 0.02, ns) * (rng.random(ns) < 0.5)
    y0[:, 1] = rng.uniform(0.5, 3.0, ns) if seed % 3 != 1 else rng.uniform(2.0, 3.2, ns)
    run_len = int(rng.choice([1, 7, 64, 512]))
    cells = synth.cell_of_system(0, ns, ncell=64, run_len=run_len)
    pr, t2m, dt, nT = synth.forcing_tables(days=days, ncell=64, seed=seed, wet_prob=float(rng.uniform(0.05, 0.5)),
                                           intensity=float(10.0 ** rng.uniform(-4.5, -3.0)), tmin=-5.0 if RADAU else 0.5, tmax=20.0)
    q = synth.minute_queries(0.0, days * 1440.0, float(rng.choice([17.0, 60.0, 240.0])))
    per_cell = bool(rng.random() < 0.5)
    block = synth.expand_per_stream(pr, t2m, cells)
    ref = O.run(O.UID_204, y0, 0.0, days * 1440.0, query=q, sp=sp, forc=block, forc_dt_h=dt, forc_nT=nT, attempt_cap=2_000_000, radau=RADAU)
    # yardstick: the same text compiled with FMA contraction.  How far it lands from `ref` on a
    # hillslope is how far two valid roundings of one computation separate there (min / max kinks:
    # Emax = min(e_pot, h_stat), the capacity overflow max(0, x1 + h_stat - Hu), ...)
    alt = O.run(O.UID_204, y0, 0.0, days * 1440.0, query=q, sp=sp, forc=block, forc_dt_h=dt, forc_nT=nT, attempt_cap=2_000_000, fma=True, radau=RADAU)
    # second yardstick: the plain build with rtol scaled by (1 +- 1e-12) -- some hillslopes sit on a
    # discrete event (the saturated static store draining through Hu, the surface store first
    # getting wet) whose step the FMA pattern happens not to move but any other rounding does
    alts = [alt] + [O.run(O.UID_204, y0, 0.0, days * 1440.0, query=q, sp=sp, forc=block, forc_dt_h=dt, forc_nT=nT, attempt_cap=2_000_000,
                          params=(0.01, 1e-6 * (1.0 + e), 1e-9, 0.9, 0.2, 10.0), radau=RADAU) for e in (1e-12, -1e-12)]
    problems = []
    for mm in (0, 1):
        eng = ShardEngine(H.Model204, y0, 0.0, days * 1440.0, sp=sp, query=q, math_mode=mm, attempt_cap=2_000_000)
        if per_cell:
            eng.set_forcings_per_cell((pr, t2m), cells, dt, nT)
        else:
            eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)
        eng.integrate(radau=RADAU)
        got = eng.results()
        flags = got["stiff"] != ref["stiff"]
        ok = ~flags & ((ref["stiff"] == 0) | (RADAU & (ref["stiff"] == 1)))
        # per hillslope: |gpu - ref| <= tolerance + 5 x (largest separation among the yardstick runs)
        over = np.zeros(ns)
        for name in ("y_final", "dense"):
            tol = 1e-9 + 1e-6 * np.abs(ref[name])
            d = (np.abs(got[name] - ref[name]) / tol).reshape(ns, -1).max(axis=1)
            spread = np.max([(np.abs(a[name] - ref[name]) / tol).reshape(ns, -1).max(axis=1) for a in alts], axis=0)
            over = np.maximum(over, d / (1.0 + 5.0 * spread))
        usable = ok & np.all([a["stiff"] == ref["stiff"] for a in alts], axis=0)
        dn = np.abs(got["n_attempt"][ok].astype(np.int64) - ref["n_attempt"][ok].astype(np.int64))
        info = dict(seed=seed, math=mm, per_cell=per_cell, run_len=run_len, flagged=int((ref["stiff"] != 0).sum()),
                    wet_end=int((ref["y_final"][:, 2] != 0).sum()), flag_mismatch=int(flags.sum()),
                    flag_mismatch_of_yardstick=int((alt["stiff"] != ref["stiff"]).sum()),
                    worst_over_bound=round(float(over[usable].max()), 3), beyond_bound=int((over[usable] > 1).sum()),
                    max_dn=int(dn.max()) if dn.size else 0, finite=bool(np.isfinite(got["y_final"]).all()))
        if flags.sum() > max(3, 3 * info["flag_mismatch_of_yardstick"]) or info["beyond_bound"] > 0.005 * ns or not info["finite"]:
            problems.append(info)
        print(info, flush=True)
    return problems


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    days = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    global RADAU
    RADAU = len(sys.argv) > 4 and sys.argv[4] == "radau"
    bad = []
    for seed in range(1000, 1000 + n):
        bad += one(seed, ns, days)
    print("FAILURES:" if bad else "all seeds within tolerance", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
