#!/usr/bin/env python3
"""Cold C3 (SURVEY's generator: T in [-5, 15]) RK45 + Radau pass, restart and continue modes, for the library given by
HLM_HIP_LIB: python tools/radau_machine_probe.py [ns] [days].  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiger_hlm_gpu_amd as H  # noqa: E402
from tiger_hlm_gpu_amd import synth  # noqa: E402
from tiger_hlm_gpu_amd.engine import ShardEngine  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
days = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sp, y0 = synth.model204_params(0, ns), synth.model204_y0(ns)
cells = synth.cell_of_system(0, ns)
pr, t2m, dt, nT = synth.forcing_tables(days=days, tmin=-5.0)
eng = ShardEngine(H.Model204, y0, 0.0, days * 1440.0, sp=sp, time_pieces=int(os.environ.get('HLM_PROBE_PIECES', '32')), queue_group_tiles=int(os.environ.get('HLM_PROBE_GROUP', '8')))
eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)
out = {"lib": os.path.basename(os.environ.get("HLM_HIP_LIB", "libhlm_hip.so")), "ns": ns, "days": days}
for tag, radau in (("rk45", False), ("restart", True), ("continue", "continue")):
    eng.integrate(radau=radau)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.integrate(radau=radau); b.record(); torch.cuda.synchronize()
    out[tag + "_ms"] = round(a.elapsed_time(b), 2)
    out[tag + "_attempts"] = eng.total_attempts()
    if radau:
        out[tag + "_unsolved"] = int((eng.stiff == 2).sum().item())
        out[tag + "_checksum"] = float(eng.y_final.double().sum().item())
print(json.dumps(out))
