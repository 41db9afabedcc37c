import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hlm():
    """The product package with its HIP library built (hipcc cross-compiles on CPU)."""
    import tiger_hlm_gpu_amd as H
    if not os.path.exists(H.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    H._lib.lib()
    return H


@pytest.fixture(scope="session")
def golden():
    g = {}
    g["g1_final"] = np.loadtxt(os.path.join(GOLDEN, "g1_dummy_final.csv"), delimiter=",", skiprows=1)
    g["g1_dense_text"] = open(os.path.join(GOLDEN, "g1_dummy_dense_sys0.csv")).read().splitlines()
    g["g2_final"] = np.load(os.path.join(GOLDEN, "g2_m204_final.npy"))
    g["g2_dense_sys0"] = np.load(os.path.join(GOLDEN, "g2_m204_dense_sys0.npy"))
    g["small_test_csv"] = os.path.join(GOLDEN, "small_test.csv")
    return g


# the solver's own tolerances: the stated HIP-vs-oracle parity bound
PARITY_ATOL = 1e-9
PARITY_RTOL = 1e-6


def assert_parity(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bound = PARITY_ATOL + PARITY_RTOL * np.abs(b)
    bad = np.abs(a - b) > bound
    assert not bad.any(), f"{what}: {bad.sum()} values exceed |d| <= 1e-9 + 1e-6|y|; max |d| = {np.abs(a - b).max():.3e}"
    return float(np.abs(a - b).max())


def fmt6(x):
    """Default std::ostream formatting of a double: 6 significant digits (%g)."""
    return "%g" % x
