#!/usr/bin/env python3
"""A/B of dense-kernel builds on ONE box (boxes of the pool differ by +-4 %, so only same-box pairs count):
    python tools/dense_variant_probe.py [--ns N] [--reps R] lib1.so lib2.so ...
Each library (a build of csrc/hlm_abi.hip with other -D flags, see tools/build_variants.sh) is loaded in a child
process (HLM_HIP_LIB) and runs the in-HBM dense kernel of bench.py: Model 204, N hillslopes x 1 day, 1-minute
queries, tile layout.  Prints ms per launch (min / mean of R) and, for -DHLM_PHASE_TIMING builds, the share of
wave cycles per loop phase.  Development aid; the product never runs this."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import tiger_hlm_gpu_amd as H
    from tiger_hlm_gpu_amd import synth
    from tiger_hlm_gpu_amd.engine import ShardEngine
    ns, days = args.ns, args.days
    tf = days * 1440.0
    q = synth.minute_queries(0.0, tf, args.stride)
    pr, t2, dt, nT = synth.forcing_tables(days=days)
    sel = None if not args.states else tuple(int(x) for x in args.states.split(","))
    e = ShardEngine(H.Model204, synth.model204_y0(ns), 0.0, tf, sp=synth.model204_params(0, ns), query=q,
                    math_mode=H.MATH_FAST, dense_tiled=args.layout, dense_states=sel)
    e.expand_forcings_on_device((pr, t2), synth.cell_of_system(0, ns), dt, nT)
    e.integrate()
    torch.cuda.synchronize()
    L = H._lib.lib()
    dbg = getattr(L, "hlm_debug_phase_cycles", None) if hasattr(L, "hlm_debug_phase_cycles") else None
    if dbg is not None:
        dbg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
        dbg(None, 1)
    ms = []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        e.integrate()
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    out = {"lib": os.path.basename(H.LIB_PATH), "ms_min": min(ms), "ms_mean": float(np.mean(ms)), "ms": ms,
           "attempts_per_hillslope": e.total_attempts() / ns,
           "checksum": float(e.dense.view(-1)[::997].sum().item())}
    width = 5 if sel is None else len(sel)
    out["TBps"] = 8.0 * width * ns * len(q) / (min(ms) * 1e-3) / 1e12
    if dbg is not None:
        arr = (C.c_ulonglong * 8)()
        dbg(arr, 0)
        tot = float(sum(arr[:4])) or 1.0
        out["phase_share"] = {"attempt": arr[0] / tot, "rows_commit": arr[1] / tot, "forcing": arr[2] / tot, "flush": arr[3] / tot}
        out["cycles_per_iteration"] = tot / max(arr[4], 1)
        out["iterations_per_wave"] = arr[4] / max(arr[5], 1)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ns", type=int, default=1_000_000)
    ap.add_argument("--days", type=int, default=1)
    ap.add_argument("--stride", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--layout", type=int, default=2)
    ap.add_argument("--states", default="")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("libs", nargs="*")
    args = ap.parse_args()
    if args.child:
        return child(args)
    libs = args.libs or [os.path.join(ROOT, "tiger_hlm_gpu_amd", "lib", "libhlm_hip.so")]
    for rnd in range(2):  # two rounds over the list: drift of the box shows as a difference between them
        for lib in libs:
            env = dict(os.environ, HLM_HIP_LIB=os.path.abspath(lib))
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--ns", str(args.ns), "--days", str(args.days),
                   "--stride", str(args.stride), "--reps", str(args.reps), "--layout", str(args.layout), "--states", args.states]
            p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            print(("round %d " % rnd) + (line[-1] if line else "FAILED %s: %s" % (lib, p.stderr[-400:])), flush=True)


if __name__ == "__main__":
    main()
