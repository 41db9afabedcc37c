"""STRICT-mode powers (csrc/hlm_strict_pow.h) on the host: the same source the device compiles, built with g++ and
checked against correctly rounded references from 60-digit decimal arithmetic -- pow(x, 0.6666666666666666) of
model_204.hpp:103 and pow(u, 0.2) of rk45_kernel.cu:151,156, with the DOUBLE exponents the reference's source spells.
glibc's pow (what the oracle and a CPU build of the reference call) is itself correctly rounded in ~999 of 1 000 cases."""
import ctypes
import decimal
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


# the header as a plain host build, and as the device build compiles it: contraction on, hardware FMAs (the compensated
# sums must survive -ffp-contract=fast: HLM_POW_OPAQUE)
@pytest.fixture(scope="module", params=["-O2 -ffp-contract=off", "-O3 -ffp-contract=fast -mfma"], ids=["contract-off", "contract-fast"])
def strict_pow(request, tmp_path_factory):
    if "-mfma" in request.param and not _cpu_has_fma():
        pytest.skip("this CPU has no FMA instructions")
    so = str(tmp_path_factory.mktemp("strict_pow") / "libstrict_pow_host.so")
    subprocess.run(["g++"] + request.param.split() + ["-shared", "-fPIC", "-I", os.path.join(ROOT, "tiger_hlm_gpu_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "strict_pow_host.cpp"), "-o", so], check=True)
    lib = ctypes.CDLL(so)

    def run(fn, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        getattr(lib, fn)(x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size))
        return out
    return run


def correctly_rounded(x, c):
    decimal.getcontext().prec = 60
    D = decimal.Decimal
    return np.array([float(D(float(v)) ** D(c)) for v in x])


@pytest.mark.parametrize("fn,c,decades", [("strict_pow_two_thirds_array", 2.0 / 3.0, [(-323, 308), (-12, 1), (-3, 3)]),
                                          ("strict_pow_fifth_array", 0.2, [(-323, 308), (0, 16), (-3, 3)]),
                                          ("strict_pow_fourth_array", 0.25, [(-323, 308), (0, 16), (-3, 3)])])
def test_strict_powers_are_correctly_rounded(strict_pow, fn, c, decades):
    rng = np.random.default_rng(7)
    wrong = total = 0
    for lo, hi in decades:   # the whole range of doubles (subnormals too), then where the solver calls them
        x = 10.0 ** rng.uniform(lo, hi, 1500)
        ref = correctly_rounded(x, c)
        got = strict_pow(fn, x)
        assert np.all(np.abs(got - ref) <= np.spacing(ref))     # never more than the neighbouring double
        wrong += int((got != ref).sum())
        total += x.size
    assert wrong <= total // 2000    # (measured: 0 of 120 000 in a longer run; glibc's pow: 16 of 20 000)
    # exactly representable results
    x = np.array([1.0, 8.0, 27.0, 0.125, 2.0 ** -300, 2.0 ** 300, 32.0, 243.0, 2.0 ** -1000, 16.0, 81.0, 2.0 ** -1074])
    ref = correctly_rounded(x, c)
    assert np.array_equal(strict_pow(fn, x), ref)


def test_strict_powers_special_values_follow_pow(strict_pow):
    libm = ctypes.CDLL("libm.so.6")
    libm.pow.restype = ctypes.c_double
    libm.pow.argtypes = [ctypes.c_double, ctypes.c_double]
    x = np.array([0.0, -0.0, -1.0, -1e-300, np.inf, -np.inf, np.nan, 5e-324, 1e-320, 2.2250738585072014e-308, 1.7976931348623157e308])
    for fn, c in (("strict_pow_two_thirds_array", 2.0 / 3.0), ("strict_pow_fifth_array", 0.2), ("strict_pow_fourth_array", 0.25)):
        want = np.array([libm.pow(float(v), c) for v in x])
        got = strict_pow(fn, x)
        assert np.array_equal(got, want, equal_nan=True)
        assert not np.signbit(got[1])   # pow(-0, c) = +0
