"""Differential fuzz of the HIP path against the oracle, bounded for the GPU suite (the generator is
tests/fuzz_parity.py's: benchmark-like but wider workloads -- capacities that do and do not saturate, wet / dry /
mixed starts, rain frequency and intensity, infiltration around the rain rates, both forcing layouts).
8 seeds x 1024 hillslopes x 3 days, both math modes, through the C ABI (ShardEngine); reference behaviour
src/solver/rk45_kernel.cu:129-163.  Every hillslope must carry the oracle's flag and be inside
|d| <= 1e-9 + 1e-6 |y| on finals and dense rows, or be individually shown ill-conditioned by the yardstick
(tests/conftest.py; exemption counts per seed and mode committed below, index sets in
tests/golden/parity_exempt_sets.json); ZERO unexplained.  Seed 1015 is round 2's one unexplained FAST hillslope
(914): it needs the yardstick's second tier, see conftest.EXTENDED_YARDSTICK_SAMPLES."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

NS, DAYS = 1024, 3
# seed: ((first-tier exemptions STRICT, FAST), (second-tier exemptions STRICT, FAST)) -- measured on MI355X (the STRICT
# column re-measured in round 4 with the correctly rounded powers of csrc/hlm_strict_pow.h: other last bits, other
# ill-conditioned hillslopes, as many of them: 4/102/66/7/23/32/13/15 before, the index sets differ)
COMMITTED = {
    1000: ((3, 5), (0, 0)),
    1001: ((102, 104), (0, 0)),
    1002: ((66, 65), (0, 0)),
    1003: ((7, 7), (0, 0)),
    1004: ((23, 21), (0, 0)),
    1005: ((32, 29), (0, 0)),
    1006: ((12, 12), (0, 0)),
    1015: ((15, 18), (0, 1)),
}


@pytest.mark.gpu
@pytest.mark.parametrize("math_mode", [0, 1])
@pytest.mark.parametrize("seed", sorted(COMMITTED))
def test_fuzz_parity_gpu(oracle, seed, math_mode):
    import fuzz_parity as F
    from conftest import (DEFAULT_PARAMS, YARDSTICK_PERTURBATIONS, assert_parity_with_exemptions, perturbed_params,
                          yardstick_runs)
    c = F.make_case(seed, NS, DAYS)
    kw = dict(query=c["q"], sp=c["sp"], forc=c["block"], forc_dt_h=c["dt"], forc_nT=c["nT"], attempt_cap=2_000_000)
    ref = oracle.run(oracle.UID_204, c["y0"], 0.0, c["tf"], **kw)
    alts = yardstick_runs(oracle, oracle.UID_204, c["y0"], 0.0, c["tf"], perturbations=YARDSTICK_PERTURBATIONS, **kw)
    got = F.gpu_run(c, math_mode)
    got_alts = [F.gpu_run(c, math_mode, perturbed_params(DEFAULT_PARAMS, er, eh)) for er, eh in YARDSTICK_PERTURBATIONS[:4]]
    assert np.isfinite(got["y_final"]).all() and np.isfinite(got["dense"]).all()
    (cap1, cap2) = COMMITTED[seed][0][math_mode], COMMITTED[seed][1][math_mode]
    n, worst = assert_parity_with_exemptions(
        got, ref, alts, cap1, what=f"fuzz seed {seed}, math {math_mode}", got_alts=got_alts,
        extended=dict(oracle=oracle, uid=oracle.UID_204, y0=c["y0"], t0=0.0, tf=c["tf"], kw=kw), max_exempt_extended=cap2)
    same = got["stiff"] == ref["stiff"]
    dn = np.abs(got["n_attempt"][same].astype(np.int64) - ref["n_attempt"][same].astype(np.int64))
    print(f"fuzz seed {seed} math {math_mode}: exempt {n} (cap {cap1}), worst within {worst:.4f}, max |d attempts| {int(dn.max())}")


@pytest.mark.gpu
@pytest.mark.parametrize("math_mode", [0, 1])
@pytest.mark.parametrize("seed", sorted(COMMITTED))
def test_fuzz_launch_shapes_agree_bit_for_bit(seed, math_mode):
    """The same fuzz cases without dense output, one wave per tile (rk45_kernel.cu:33) against the queue kernel
    (hlm_device_job.time_pieces, csrc/hlm_rk45_queue_kernel.h) with a piece count drawn from the seed (2 ... 14, so
    that pieces end inside forcing hours as well as on them) and with 72 pieces (one per forcing hour): finals, flags
    and both step counters of every hillslope identical -- 16 tiles on 3 072 resident waves, so every piece but the
    first waits for another wave's."""
    import fuzz_parity as F
    c = F.make_case(seed, NS, DAYS)
    a = F.gpu_run(c, math_mode, time_pieces=0)
    for pieces in (2 + seed % 13, 72):
        b = F.gpu_run(c, math_mode, time_pieces=pieces)
        for k in ("y_final", "stiff", "n_attempt", "n_accept"):
            assert np.array_equal(a[k], b[k]), (seed, math_mode, pieces, k)
