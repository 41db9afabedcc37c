"""Timing probe for the Radau-IIA fallback: a cold-cell workload (T in [-5, 15]: the negative ET
stub flags ~18 % of the hillslopes in RK45), RK45 pass alone vs RK45 + Radau pass.
Run on the GPU box:  python tools/radau_probe.py [ns] [days]   (under rocprofv3 for kernel stats)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiger_hlm_gpu_amd as H  # noqa: E402
from tiger_hlm_gpu_amd import synth  # noqa: E402
from tiger_hlm_gpu_amd.engine import ShardEngine  # noqa: E402


def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    days = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    sp = synth.model204_params(0, ns)
    y0 = synth.model204_y0(ns)
    cells = synth.cell_of_system(0, ns, run_len=64)
    pr, t2m, dt, nT = synth.forcing_tables(days=days, tmin=-5.0)
    eng = ShardEngine(H.Model204, y0, 0.0, days * 1440.0, sp=sp)
    eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)

    def timed(radau):
        torch.cuda.synchronize()
        t = time.perf_counter()
        eng.integrate(radau=radau)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3

    timed(True)  # warm-up (first hipMallocAsync creates the pool)
    rk = min(timed(False) for _ in range(3))
    flagged = int((eng.stiff == 1).sum())
    att_rk = eng.total_attempts()
    both = min(timed(True) for _ in range(3))
    st = eng.stiff.cpu().numpy()
    print(json.dumps({"ns": ns, "days": days, "flagged": flagged, "rk45_ms": round(rk, 2), "rk45_plus_radau_ms": round(both, 2),
                      "radau_attempts": eng.total_attempts() - att_rk, "unsolved_after": int((st == 2).sum()),
                      "finite": bool(torch.isfinite(eng.y_final).all())}))


if __name__ == "__main__":
    main()
