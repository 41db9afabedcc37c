#!/usr/bin/env python3
"""STRICT powers on the device against 60-digit decimal references: where do wrong roundings come from?  Prints the
seed residuals (hlm_selftest_math 7 / 8) and the wrong roundings by argument range."""
import decimal
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import tiger_hlm_gpu_amd as H  # noqa: E402

L = H._lib.lib()
decimal.getcontext().prec = 60


def run(which, x):
    xin = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    out = torch.empty_like(xin)
    H._lib.check(L.hlm_selftest_math(which, xin.data_ptr(), out.data_ptr(), xin.numel(), None))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def cr(v, c):
    return np.array([float(decimal.Decimal(float(a)) ** decimal.Decimal(c)) for a in v])


rng = np.random.default_rng(3)
for name, which, wres, c, ranges in (("x^(2/3)", 3, 7, 2.0 / 3.0, [(0, 3), (-40, 40), (-1000, 1000)]), ("u^0.2", 4, 8, 0.2, [(0, 5), (-40, 54), (-1000, 1000)])):
    for lo, hi in ranges:
        x = 2.0 ** rng.uniform(lo, hi, 20000)
        arg = x if which == 3 else 1.0 / x          # 4 evaluates pow(1.0 / x, 0.2)
        got = run(which, x)
        ref = cr(arg, c)
        e = run(wres, arg)
        bad = got != ref
        print(f"{name} on 2^[{lo},{hi}]: wrong roundings {bad.mean():.4f}, max |ulp| {np.abs((got - ref) / np.spacing(ref)).max():.1f}, "
              f"seed residual |e| max {np.abs(e).max():.3e} median {np.median(np.abs(e)):.3e}; |e| among the wrong: median {np.median(np.abs(e[bad])) if bad.any() else 0:.3e}", flush=True)
