#!/usr/bin/env python3
"""Where the waves of a queue launch spend their time (experiment build -DHLM_QUEUE_PROF via HLM_HIP_LIB): per schedule
(HLM_QUEUE_RUNS="16,8,4,2,2" or queue_run_pieces) the launch's ms, and over the resident waves: body / wait / idle-at-the-end
ticks, items, items that ran.   python tools/queue_prof_probe.py [--tmin -5] [--ns N] --sched 1 auto 16,8,4,2,2 ..."""
import argparse
import ctypes as C
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
ap.add_argument("--pieces", type=int, default=32)
ap.add_argument("--sched", nargs="*", default=["1", "auto"])
ap.add_argument("--group", type=int, default=0, help="hlm_device_job.queue_group_tiles")
a = ap.parse_args()
L = H._lib.lib()
assert hasattr(L, "hlm_debug_queue_prof"), "needs a -DHLM_QUEUE_PROF build (HLM_HIP_LIB)"
sp = synth.model204_params(0, a.ns)
y0 = synth.model204_y0(a.ns)
cells = synth.cell_of_system(0, a.ns)
pr, t2m, dt, nT = synth.forcing_tables(days=a.days, tmin=a.tmin)
tf = a.days * 1440.0
NB = 4096
prof = torch.zeros(NB * 8, dtype=torch.int64, device="cuda")
for sched in a.sched:
    os.environ.pop("HLM_QUEUE_RUNS", None)
    rp = 0
    if sched == "auto":
        pass
    elif "," in sched:
        os.environ["HLM_QUEUE_RUNS"] = sched
    else:
        rp = int(sched)
    eng = ShardEngine(H.Model204, y0, 0.0, tf, sp=sp, math_mode=H.MATH_FAST, time_pieces=a.pieces, queue_run_pieces=rp, queue_group_tiles=a.group)
    eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)
    L.hlm_debug_queue_prof(C.c_void_p(0))
    eng.integrate()
    torch.cuda.synchronize()
    ms = []
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.integrate(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    prof.zero_()
    L.hlm_debug_queue_prof(C.c_void_p(prof.data_ptr()))
    eng.integrate()
    torch.cuda.synchronize()
    L.hlm_debug_queue_prof(C.c_void_p(0))
    p = prof.cpu().numpy().reshape(NB, 8)
    p = p[p[:, 4] > 0]
    t0, t1 = p[:, 0].min(), p[:, 1].max()
    span = (t1 - t0) / 1e5   # ms (100 MHz ticks)
    body, wait = p[:, 2] / 1e5, p[:, 3] / 1e5
    tail = (t1 - p[:, 1]) / 1e5
    print(f"sched {sched:>14s}: {np.mean(ms):8.3f} ms | waves {len(p)}, span {span:7.2f} ms, per wave: body {body.mean():7.2f} (min {body.min():.2f} max {body.max():.2f}) "
          f"wait {wait.mean():6.3f} (max {wait.max():.2f}) idle at end {tail.mean():5.2f} (max {tail.max():.2f}) | items/wave {p[:, 4].mean():.1f}, ran {p[:, 5].mean():.1f}, "
          f"lanes per item that ran {(p[:, 6] & ((1 << 40) - 1)).sum() / max(p[:, 5].sum(), 1):.1f}, lanes per trip of the attempt loop "
          f"{(p[:, 7] & ((1 << 40) - 1)).sum() / max((p[:, 7] >> 40).sum(), 1):.1f}, trips with a forcing refresh "
          f"{(p[:, 6] >> 40).sum() / max((p[:, 7] >> 40).sum(), 1):.3f}", flush=True)
    del eng
    torch.cuda.empty_cache()
