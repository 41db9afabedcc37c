"""The Radau-IIA fallback's CPU restatement (oracle/hlm_oracle_radau.c) against independent
solutions: SciPy's Radau at tight tolerance, the matrix exponential of the linear DummyModel,
and a piecewise SciPy integration over the forcing samples.  No reference artifact exists for
this row (the reference never copies its Radau results back) -- parity for it is
"validated against SciPy", not pinned.  CPU only.

Errors are in units of the solver tolerance: |d| / (1e-9 + 1e-6 |y|)."""
import numpy as np
from scipy.integrate import solve_ivp
from scipy.linalg import expm

from tiger_hlm_gpu_amd import synth


def _scaled(a, b):
    return np.abs(a - b) / (1e-9 + 1e-6 * np.abs(b))


def test_radau_dummy_linear_system_vs_matrix_exponential(oracle):
    """DummyModel is y' = A y + b: the exact flow is known."""
    A = np.array([[-0.5, 0, 0, 0, 0], [0.5, -0.9, 0, 0, 0], [0, 0.3, 0, 0, 0], [0, 0.6, 0, -0.4, 0], [0, 0, 0, 0.4, 0]])
    b = np.array([1.0, 0.8, -0.2, -0.3, -0.1])
    y0 = synth.dummy_y0(0, 6)
    # the model really is that affine map
    for y in y0:
        assert np.allclose(oracle.rhs(oracle.UID_DUMMY, y), A @ y + b, rtol=0, atol=1e-15)
    q = synth.dummy_query_times(5.0, 50)
    r = oracle.run(oracle.UID_DUMMY, y0, 0.0, 5.0, query=q, radau_only=True)
    assert (r["stiff"] == 1).all()  # success leaves the RK45 flag in place

    def exact(y, t):  # augmented system [[A, b], [0, 0]]
        M = np.zeros((6, 6))
        M[:5, :5], M[:5, 5] = A, b
        return (expm(M * t) @ np.append(y, 1.0))[:5]

    for s in range(6):
        assert _scaled(r["y_final"][s], exact(y0[s], 5.0)).max() <= 1.0
        for k in (0, 17, 49):
            assert _scaled(r["dense"][s, k], exact(y0[s], q[k])).max() <= 1.0
    assert r["n_attempt"].max() <= 40  # a smooth linear problem: a couple of dozen implicit steps


def test_radau_model204_constant_forcing_vs_scipy_radau(oracle):
    ns = 4
    sp = synth.model204_params(0, ns, seed=3)
    y0 = synth.model204_y0(ns, wet_surface=True)
    stub = np.array([np.float32(2e-5), np.float32(8.0)], dtype=np.float64)
    q = synth.minute_queries(0, 2880.0, 60.0)
    r = oracle.run(oracle.UID_204, y0, 0.0, 2880.0, query=q, sp=sp, forc_stub_f64=stub, radau_only=True)
    k = oracle.run(oracle.UID_204, y0, 0.0, 2880.0, sp=sp, forc_stub_f64=stub)
    assert (r["n_attempt"] < k["n_attempt"] / 3).all()  # mildly stiff: far fewer implicit steps
    for s in range(ns):
        f = lambda t, y: oracle.rhs(oracle.UID_204, y, sp[s:s + 1], stub.astype(np.float32), t)
        ref = solve_ivp(f, (0, 2880.0), y0[s], method="Radau", rtol=1e-11, atol=1e-14, t_eval=q[q > 0])
        assert _scaled(r["y_final"][s], ref.y[:, -1]).max() <= 1.0
        assert _scaled(r["dense"][s][q > 0], ref.y.T).max() <= 1.0
        assert np.all(r["dense"][s][q <= 0] == 0.0)  # queries at t0 are never produced


def test_radau_rescues_flagged_systems_and_matches_piecewise_scipy(oracle):
    """Cold cells flag ~1/6 of these hillslopes in RK45; the fallback finishes all of them, leaves
    the others untouched, and agrees with an independent integration that restarts SciPy's Radau
    at every forcing sample (which is what clipping the implicit steps to the sample boundaries
    amounts to)."""
    ns, days = 128, 5
    sp = synth.model204_params(0, ns, seed=3)
    y0 = synth.model204_y0(ns)
    pr, t2m, dt, nT = synth.forcing_tables(days=days, ncell=16, seed=5, tmin=-5.0)
    cells = synth.cell_of_system(0, ns, ncell=16, run_len=16)
    forc = synth.expand_per_stream(pr, t2m, cells)
    tf = days * 1440.0
    k = oracle.run(oracle.UID_204, y0, 0.0, tf, sp=sp, forc=forc, forc_dt_h=dt, forc_nT=nT)
    r = oracle.run(oracle.UID_204, y0, 0.0, tf, sp=sp, forc=forc, forc_dt_h=dt, forc_nT=nT, radau=True)
    fl = k["stiff"] == 1
    assert 0 < fl.sum() < ns
    assert np.array_equal(r["stiff"], k["stiff"])
    assert np.all(k["y_final"][fl] == 0.0) and np.all(np.isfinite(r["y_final"]))
    assert np.all(np.abs(r["y_final"][fl]).sum(axis=1) > 0)
    assert np.array_equal(r["y_final"][~fl], k["y_final"][~fl])
    assert np.array_equal(r["n_attempt"][~fl], k["n_attempt"][~fl])
    assert (r["t_reached"][fl] == tf).all()
    s = int(np.nonzero(fl)[0][0])
    y = y0[s].copy()
    for hr in range(24 * days):
        F = np.array([pr[hr, cells[s]], t2m[hr // 24, cells[s]]], dtype=np.float32)
        sol = solve_ivp(lambda t, yy: oracle.rhs(oracle.UID_204, yy, sp[s:s + 1], F, t), (hr * 60.0, (hr + 1) * 60.0), y,
                        method="Radau", rtol=1e-9, atol=1e-12)
        y = sol.y[:, -1]
    assert _scaled(r["y_final"][s], y).max() <= 5.0  # global error over 120 restarts, kink at Hu included


def test_radau_continue_from_the_flag_state_matches_piecewise_scipy(oracle):
    """radau_continue (build option; the reference's kernel restarts at t0, radau_kernel.cu:44-52): a flagged
    hillslope keeps the rows RK45 wrote up to the state it reached and the implicit method takes over from there.
    Checked against SciPy's Radau started at that state and restarted at every forcing sample."""
    ns, days = 128, 5
    sp = synth.model204_params(0, ns, seed=3)
    y0 = synth.model204_y0(ns)
    pr, t2m, dt, nT = synth.forcing_tables(days=days, ncell=16, seed=5, tmin=-5.0)
    cells = synth.cell_of_system(0, ns, ncell=16, run_len=16)
    forc = synth.expand_per_stream(pr, t2m, cells)
    tf = days * 1440.0
    q = np.arange(30.0, tf + 1.0, 60.0)
    kw = dict(sp=sp, forc=forc, forc_dt_h=dt, forc_nT=nT, query=q)
    k = oracle.run(oracle.UID_204, y0, 0.0, tf, **kw)
    r = oracle.run(oracle.UID_204, y0, 0.0, tf, radau=True, radau_continue=True, **kw)
    full = oracle.run(oracle.UID_204, y0, 0.0, tf, radau=True, **kw)
    fl = k["stiff"] == 1
    assert 0 < fl.sum() < ns and np.array_equal(r["stiff"], k["stiff"])
    assert np.array_equal(r["y_final"][~fl], k["y_final"][~fl]) and np.array_equal(r["dense"][~fl], k["dense"][~fl])
    assert (r["t_reached"][fl] == tf).all() and np.all(np.isfinite(r["y_final"]))
    assert (r["n_attempt"][fl] < full["n_attempt"][fl]).all()   # nothing before the flag is integrated twice
    for s in np.nonzero(fl)[0][:4]:
        t_flag = k["t_reached"][s]
        before = q <= t_flag
        assert np.array_equal(r["dense"][s][before], k["dense"][s][before])      # RK45's rows stay
        assert np.all(np.abs(r["dense"][s][~before]).sum(axis=1) > 0)              # the later ones are filled
        y, t = k["y_reached"][s].copy(), t_flag
        while t < tf:
            hr = min(int(t // 60.0), 24 * days - 1)
            t1 = min((hr + 1) * 60.0, tf)
            F = np.array([pr[hr, cells[s]], t2m[min(hr // 24, days - 1), cells[s]]], dtype=np.float32)
            sol = solve_ivp(lambda tt, yy: oracle.rhs(oracle.UID_204, yy, sp[s:s + 1], F, tt), (t, t1), y, method="Radau", rtol=1e-9, atol=1e-12)
            y, t = sol.y[:, -1], t1
        assert _scaled(r["y_final"][s], y).max() <= 5.0


def test_radau_attempt_cap_flags_and_zeroes(oracle):
    y0 = synth.dummy_y0(0, 3)
    r = oracle.run(oracle.UID_DUMMY, y0, 0.0, 5.0, query=synth.dummy_query_times(5.0, 20), radau_only=True, attempt_cap=4)
    assert (r["stiff"] == 2).all() and np.all(r["y_final"] == 0.0) and (r["n_attempt"] == 4).all()
    assert np.all(r["dense"][:, -1] == 0.0)  # rows past the cap stay zero


def test_analytic_jacobian_equals_finite_differences(oracle):
    """The Radau pass takes d(rhs)/dy from the model instead of the reference's finite differences
    (approx_jacobian, radau_step_dense.cuh:13-31).  Away from the kinks of the min / max terms the two
    must agree: random Model 204 states in every regime (cold / warm, saturated / not, wet / dry /
    negative surface store, infiltration and percolation limited or not) and the DummyModel."""
    from tiger_hlm_gpu_amd import synth
    rng = np.random.default_rng(5)
    n = 400
    sp = synth.model204_params(0, n, seed=11)
    sp["Hu"] = np.where(rng.random(n) < 0.5, sp["Hu"], 10.0 ** rng.uniform(-3, 0, n))
    worst = 0.0
    checked = 0
    for i in range(n):
        y = np.array([rng.uniform(0, 0.05), rng.uniform(0, 3.0), rng.choice([0.0, rng.uniform(1e-6, 0.02), -1e-4]),
                      rng.uniform(0, 5), rng.uniform(0, 1)])
        F = (rng.choice([0.0, rng.uniform(1e-6, 1e-3)]), rng.uniform(-5, 15))
        Ja = oracle.jacobian(oracle.UID_204, y, sp[i:i + 1], F)
        f0 = oracle.rhs(oracle.UID_204, y, sp[i:i + 1], F)
        Jf = np.zeros((5, 5))
        kink = False
        for j in range(5):
            d = 1e-7 * max(1.0, abs(y[j]))
            yp, ym = y.copy(), y.copy()
            yp[j] += d
            ym[j] -= d
            fp, fm = oracle.rhs(oracle.UID_204, yp, sp[i:i + 1], F), oracle.rhs(oracle.UID_204, ym, sp[i:i + 1], F)
            Jf[:, j] = (fp - fm) / (2 * d)
            # a kink inside [y - d, y + d]: the one-sided slopes differ, the state is skipped
            kink |= bool(np.any(np.abs((fp - f0) - (f0 - fm)) > 1e-6 * d + 1e-18)) and y[2] <= 0.0 or \
                bool(np.any(np.abs((fp - f0) / d - (f0 - fm) / d) > 1e-3 * (np.abs(Jf[:, j]) + 1e-12)))
        if kink:
            continue
        checked += 1
        worst = max(worst, float(np.abs(Ja - Jf).max() / (np.abs(Jf).max() + 1e-12)))
    assert checked > n // 2 and worst < 1e-5, (checked, worst)
    M = np.array([[-0.5, 0, 0, 0, 0], [0.5, -0.9, 0, 0, 0], [0, 0.3, 0, 0, 0], [0, 0.6, 0, -0.4, 0], [0, 0, 0, 0.4, 0]])
    assert np.array_equal(oracle.jacobian(oracle.UID_DUMMY, np.ones(5)), M)


def test_radau_pass_from_given_flag_states_equals_the_continue_mode(oracle):
    """oracle.run(radau_only=True, continue_from=(t, y)): the implicit pass alone, taking every system over at a GIVEN
    state -- with the states the oracle's own RK45 run reached when it flagged them it must reproduce
    radau=True, radau_continue=True bit for bit (the GPU test hands it the GPU's flag states instead)."""
    from tiger_hlm_gpu_amd import synth
    O = oracle
    ns, days = 96, 6
    sp, y0 = synth.model204_params(0, ns), synth.model204_y0(ns)
    cells = synth.cell_of_system(0, ns, run_len=64)
    pr, t2m, dt, nT = synth.forcing_tables(days=days, tmin=-5.0)
    tf = days * 1440.0
    q = synth.minute_queries(0.0, tf, 45.0)
    kw = dict(sp=sp, forc=synth.expand_per_stream(pr, t2m, cells), forc_dt_h=dt, forc_nT=nT)
    plain = O.run(O.UID_204, y0, 0.0, tf, query=q, **kw)
    ref = O.run(O.UID_204, y0, 0.0, tf, query=q, radau=True, radau_continue=True, **kw)
    idx = np.flatnonzero(plain["stiff"] == 1)
    assert 5 < len(idx) < ns
    own = O.run(O.UID_204, y0[idx], 0.0, tf, query=q, sp=sp[idx], forc=kw["forc"][:, idx], forc_dt_h=dt, forc_nT=nT, radau_only=True,
                continue_from=(plain["t_reached"][idx], plain["y_reached"][idx]))
    assert np.array_equal(own["y_final"], ref["y_final"][idx])
    after = q[None, :] > plain["t_reached"][idx][:, None]
    assert np.array_equal(own["dense"][after], ref["dense"][idx][after])


def test_radau_attempt_trace_accounts_for_every_attempt(oracle):
    """hlm_oracle_job.radau_trace (the diagnostic behind tests/radau_wave_sim.py): one byte per implicit attempt -- as many as
    n_attempt counts, as many accepted ones as n_accept, every attempt with 1 ... 10 Newton iterations, and a landing on a
    forcing boundary for every sample of the finest stream but the last (the final step lands on tf instead)."""
    from tiger_hlm_gpu_amd import synth
    ns, days = 6, 3
    sp, y0 = synth.model204_params(0, ns), synth.model204_y0(ns)
    cells = synth.cell_of_system(0, ns, run_len=2)
    pr, t2m, dt, nT = synth.forcing_tables(days=days, tmin=-5.0)
    r = oracle.run(oracle.UID_204, y0, 0.0, days * 1440.0, sp=sp, forc=synth.expand_per_stream(pr, t2m, cells), forc_dt_h=dt,
                   forc_nT=nT, radau_only=True, radau_trace_cap=20000)
    plain = oracle.run(oracle.UID_204, y0, 0.0, days * 1440.0, sp=sp, forc=synth.expand_per_stream(pr, t2m, cells), forc_dt_h=dt,
                       forc_nT=nT, radau_only=True)
    assert np.array_equal(r["y_final"], plain["y_final"]) and plain["radau_trace"] is None
    for s in range(ns):
        n = int(r["n_attempt"][s])
        b = r["radau_trace"][s]
        assert 0 < n < 20000 and np.all(b[n:] == 0) and np.all((b[:n] & 15) >= 1) and np.all((b[:n] & 15) <= 10)
        accepted = (b[:n] & 0x60) == 0
        assert accepted.sum() == r["n_accept"][s]
        assert ((b[:n] & 0x10) != 0).sum() == 24 * days - 1
