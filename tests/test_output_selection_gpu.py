"""Device-side output selection (ABI 3): hlm_device_job.dense_ncomp / hlm_set_output_states -- the reference's
``output.states`` (data/config.yaml:44-52, parsed by I_O/config_loader.cpp:19-84; the writers take the variable
list, I_O/output_series.cpp:18-72).  The dense rows the kernels store, the transposing kernel moves and PCIe
carries hold the K selected components only.  Checked on the GPU, through the C ABI:
  * against the oracle's rows restricted to the subset (parity bound |d| <= 1e-9 + 1e-6 |y|),
  * bit for bit against the columns of the unselected run (selection changes no step),
in the three device layouts, streamed in query windows, after the Radau pass, through the host API and the driver."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_parity

pytestmark = pytest.mark.gpu

SUBSETS = [(1,), (0, 2), (1, 3, 4), (0, 1, 2, 3), (4,)]


@pytest.fixture(scope="module")
def gpu(hlm):
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return hlm


def _case(H, ns=300, days=2, wet=True, tmin=0.5):
    from tiger_hlm_gpu_amd import synth
    sp = synth.model204_params(0, ns)
    y0 = synth.model204_y0(ns, wet_surface=wet)
    cells = synth.cell_of_system(0, ns, run_len=64)
    pr, t2m, dt, nT = synth.forcing_tables(days=days, tmin=tmin)
    q = np.concatenate([[-5.0, 0.0], synth.minute_queries(1.0, days * 1440.0 + 30.0, 7.0)])  # before t0 .. past tf
    return sp, y0, cells, pr, t2m, dt, nT, q


def _engine(H, case, days, **kw):
    from tiger_hlm_gpu_amd.engine import ShardEngine
    sp, y0, cells, pr, t2m, dt, nT, q = case
    eng = ShardEngine(H.Model204, y0, 0.0, days * 1440.0, sp=sp, query=q, **kw)
    eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)
    return eng


@pytest.mark.parametrize("math_mode", [0, 1])
@pytest.mark.parametrize("layout", [0, 2])
def test_selected_rows_in_every_device_layout_vs_oracle_and_vs_the_full_run(gpu, oracle, layout, math_mode):
    H = gpu
    from tiger_hlm_gpu_amd import synth
    days = 2
    case = _case(H, ns=300, days=days)  # 300 = 4 full tiles + a ragged one
    sp, y0, cells, pr, t2m, dt, nT, q = case
    ref = oracle.run(oracle.UID_204, y0, 0.0, days * 1440.0, query=q, sp=sp, forc=synth.expand_per_stream(pr, t2m, cells),
                     forc_dt_h=dt, forc_nT=nT)
    full_eng = _engine(H, case, days, dense_tiled=layout, math_mode=math_mode)
    full_eng.integrate()
    full = full_eng.results()
    for sel in SUBSETS:
        eng = _engine(H, case, days, dense_tiled=layout, math_mode=math_mode, dense_states=sel)
        assert eng.dense.shape[-1] == len(sel)
        eng.integrate()
        got = eng.results()
        assert got["dense"].shape == (300, len(q), len(sel))
        assert np.array_equal(got["dense"], full["dense"][:, :, list(sel)]), (layout, sel)  # same steps, same rows
        for k in ("y_final", "stiff", "n_attempt", "n_accept"):
            assert np.array_equal(got[k], full[k]), (k, sel)
        assert_parity(got["dense"], ref["dense"][:, :, list(sel)], f"selection {sel}, layout {layout}")
    # listing every component is the plain kernel
    eng = _engine(H, case, days, dense_tiled=layout, math_mode=math_mode, dense_states=(0, 1, 2, 3, 4))
    eng.integrate()
    assert np.array_equal(eng.results()["dense"], full["dense"])


@pytest.mark.parametrize("path", ["copy-engine", "transfer-kernel", "pageable"])
@pytest.mark.parametrize("window", [1, 37, 1000])
def test_selected_rows_streamed_in_windows_equal_one_shot(gpu, window, path):
    H = gpu
    days = 2
    case = _case(H, ns=300, days=days)
    q = case[-1]
    full = _engine(H, case, days)
    full.integrate()
    one = full.results()
    for sel in ((2,), (0, 3), (0, 1, 3, 4)):
        eng = _engine(H, case, days, dense_states=sel)
        H._lib.lib().hlm_set_stream_path({"transfer-kernel": 2}.get(path, 0))
        try:
            host_out = np.full((300, len(q), len(sel)), np.nan) if path == "pageable" else None
            host, stats = eng.integrate_streamed(window, host_out=host_out)
        finally:
            H._lib.lib().hlm_set_stream_path(0)
        assert stats["dense_bytes"] == 300 * len(q) * 8 * len(sel)   # K/5 of the bytes cross PCIe
        assert np.array_equal(host, one["dense"][:, :, list(sel)]), (sel, window, path)
        two = eng.results()
        for k in ("y_final", "stiff", "n_attempt", "n_accept"):
            assert np.array_equal(one[k], two[k]), k


@pytest.mark.parametrize("layout", [0, 2])
def test_selection_with_flagged_hillslopes_and_the_radau_pass(gpu, layout):
    """Cold cells: flagged hillslopes have zero rows (RK45 pass) or the implicit pass's rows (Radau) -- in both
    cases the selected columns of the full run."""
    H = gpu
    days = 8
    case = _case(H, ns=256, days=days, wet=False, tmin=-5.0)
    for radau in (False, True):
        full = _engine(H, case, days, dense_tiled=layout)
        full.integrate(radau=radau)
        one = full.results()
        assert 0 < (one["stiff"] == 1).sum() < 256
        for sel in ((1,), (0, 2, 4)):
            eng = _engine(H, case, days, dense_tiled=layout, dense_states=sel)
            eng.integrate(radau=radau)
            got = eng.results()
            assert np.array_equal(got["dense"], one["dense"][:, :, list(sel)]), (radau, sel)
            assert np.array_equal(got["stiff"], one["stiff"]) and np.array_equal(got["y_final"], one["y_final"])


def test_selection_errors(gpu):
    H = gpu
    case = _case(H, ns=64, days=1)
    for bad in ((5,), (2, 1), (1, 1), (-1,), (0, 1, 2, 3, 4, 5)):
        eng = _engine(H, case, 1, dense_states=bad)
        with pytest.raises(H.HlmError) as e:
            eng.integrate()
        assert e.value.code == 1, bad
    with pytest.raises(H.HlmError):
        H.api.set_output_states((3, 2))


def test_host_api_output_states(gpu):
    """hlm_set_output_states -> setup_gpu_buffers sizes the device rows for K components, the kernels store only
    those, retrieve_and_free / run_rk45 deliver h_dense[(s*nq + q)*K + k] (toy model too: N_EQ = 3)."""
    H = gpu
    from tiger_hlm_gpu_amd import synth
    api = H.api
    ns, days = 130, 1
    sp, y0, cells, pr, t2m, dt, nT, q = _case(H, ns=ns, days=days)
    d_sp = api.upload_spatial_params(sp)
    f = api.set_forcings(np.concatenate([pr, t2m], axis=0), dt, nT, pr.shape[1], col=cells)
    try:
        api.set_output_states(None)
        full_final, full_dense = api.run_rk45(H.Model204, y0, 0.0, days * 1440.0, q, d_sp)
        full_dense = full_dense.reshape(ns, len(q), 5)
        api.set_output_states((1, 4))
        final, dense = api.run_rk45(H.Model204, y0, 0.0, days * 1440.0, q, d_sp, dense_width=2)
        assert np.array_equal(final, full_final)
        assert np.array_equal(dense.reshape(ns, len(q), 2), full_dense[:, :, [1, 4]])
        # the three-phase form, into pinned memory (the transposing kernel writes it directly)
        import torch
        pinned = torch.empty(ns * len(q) * 2, dtype=torch.float64).pin_memory()
        bufs = api.setup_gpu_buffers(H.Model204, y0, q)
        assert bufs.dense_ncomp == 2 and list(bufs.dense_comp)[:2] == [1, 4]
        api.launch_rk45_kernel(H.Model204, bufs, 0.0, days * 1440.0, d_sp)
        _, d2 = api.retrieve_and_free(H.Model204, bufs, dense_out=pinned)
        assert np.array_equal(d2.reshape(ns, len(q), 2), full_dense[:, :, [1, 4]])
        api.set_output_states((0, 1, 2, 3))      # 4 of 5
        _, d4 = api.run_rk45(H.Model204, y0, 0.0, days * 1440.0, q, d_sp, dense_width=4)
        assert np.array_equal(d4.reshape(ns, len(q), 4), full_dense[:, :, :4])
        api.set_output_states((4,))              # not a state of the 3-state toy model
        with pytest.raises(H.HlmError):
            api.run_rk45(H.ToyModel3, np.ones((8, 3)), 0.0, 10.0, np.arange(1.0, 10.0), None, dense_width=1)
    finally:
        api.set_output_states(None)
        f.free()
        d_sp.free()


def test_cpp_driver_states_selected_on_the_device(gpu, golden, tmp_path):
    """rk45_solver --out-format nc --states 1,3: the selection is made on the device (dense_row_width 2 in the
    result line) and the NetCDF files equal those of the host-side subset of a csv+nc run (--out-format both)."""
    exe = os.path.join(ROOT, "tools", "rk45_solver")
    h5dump = os.path.join(os.environ.get("HDF5_DIR", "/opt/conda"), "bin", "h5dump")
    if not os.path.exists(exe) or not os.path.exists(h5dump):
        pytest.skip("driver or h5dump not available")
    base = [exe, "--model", "204", "--params", golden["small_test_csv"], "--tf", "2880", "--query-stride", "30", "--initial-step", "1e-6",
            "--states", "1,3"]
    outs = {}
    for fmt in ("nc", "both"):
        d = tmp_path / fmt
        d.mkdir()
        p = subprocess.run(base + ["--out-format", fmt, "--out-final", str(d / "final.csv"), "--out-dense", str(d / "dense.csv")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        line = json.loads(p.stdout.strip().splitlines()[-1])
        assert line["device_output_selection"] is (fmt == "nc") and line["dense_row_width"] == (2 if fmt == "nc" else 5)
        assert set(line["phases"]) == {"params_csv_s", "forcing_read_s", "h2d_s", "integrate_s", "d2h_s", "write_s", "window_sink_s"}
        vals = []
        for name in ("final", "dense"):
            for ds in ("/outputs", "/variable"):
                b = str(d / f"{name}{ds.replace('/', '_')}.bin")
                subprocess.run([h5dump, "-d", ds, "-b", "LE", "-o", b, str(d / f"{name}.nc")], check=True, capture_output=True)
                vals.append(open(b, "rb").read())
        outs[fmt] = vals
    assert outs["nc"] == outs["both"]
    dense = np.frombuffer(outs["nc"][2], dtype=np.float64)
    assert dense.size == 10 * 97 * 2 and np.frombuffer(outs["nc"][3], dtype=np.int32).tolist() == [1, 3]


@pytest.mark.parametrize("window", [1, 37, 1000])
def test_streamed_sink_delivers_every_window_once_in_order(gpu, window):
    """hlm_rk45_integrate_streamed_sink: the rows pass through the library's two pinned staging buffers into the
    consumer's callback, window by window, [ns][W][K]; assembled they equal the one-shot run bit for bit (all
    components and a device-side selection), and the integrator state at the end is the one-shot run's."""
    H = gpu
    days = 2
    case = _case(H, ns=300, days=days)
    q = case[-1]
    full = _engine(H, case, days)
    full.integrate()
    one = full.results()
    for sel in (None, (0, 3)):
        eng = _engine(H, case, days, dense_states=sel)
        width = 5 if sel is None else len(sel)
        out = np.full((300, len(q), width), np.nan)
        calls = []

        def sink(qb, qe, rows):
            assert rows.shape == (300, qe - qb, width)
            calls.append((qb, qe))
            out[:, qb:qe] = rows          # (a copy: the view is only valid until the sink returns)
        st = eng.integrate_streamed_sink(window, sink)
        W = min(window, len(q))
        assert calls == [(k * W, min((k + 1) * W, len(q))) for k in range(-(-len(q) // W))]
        assert st["windows"] == len(calls) and st["dense_bytes"] == 300 * len(q) * 8 * width
        want = one["dense"] if sel is None else one["dense"][:, :, list(sel)]
        assert np.array_equal(out, want)
        two = eng.results()
        for k in ("y_final", "stiff", "n_attempt", "n_accept"):
            assert np.array_equal(one[k], two[k]), k
    # no consumer: the rows are discarded, the statistics still count them
    st = _engine(H, case, days).integrate_streamed_sink(64, None)
    assert st["dense_bytes"] == 300 * len(q) * 40 and st["windows"] == -(-len(q) // 64)


def test_streamed_sink_stops_when_the_consumer_says_so(gpu):
    H = gpu
    case = _case(H, ns=128, days=1)
    eng = _engine(H, case, 1)
    seen = []

    def stop_after_two(qb, qe, rows):
        seen.append(qb)
        return 7 if len(seen) == 2 else 0
    with pytest.raises(H.HlmError) as e:
        eng.integrate_streamed_sink(16, stop_after_two)
    assert e.value.code == 5 and "7" in str(e.value) and len(seen) == 2

    def broken(qb, qe, rows):
        raise KeyError("consumer bug")
    with pytest.raises(KeyError):
        eng.integrate_streamed_sink(16, broken)
    eng.integrate()                       # the library is usable afterwards
    assert np.isfinite(eng.results()["y_final"]).all()


def test_host_api_streamed_run_and_the_driver_stream_dense(gpu, golden, tmp_path):
    """hlm_run_rk45_streamed (run_rk45 whose dense rows go to a consumer window by window) with the registered
    forcings and output selection, and rk45_solver --stream-dense W: dense.nc written one hyperslab per window
    while the next window integrates, equal to the file of the run that held every row."""
    H = gpu
    api = H.api
    ns, days = 130, 1
    sp, y0, cells, pr, t2m, dt, nT, q = _case(H, ns=ns, days=days)
    d_sp = api.upload_spatial_params(sp)
    f = api.set_forcings(np.concatenate([pr, t2m], axis=0), dt, nT, pr.shape[1], col=cells)
    try:
        api.set_output_states(None)
        full_final, full_dense = api.run_rk45(H.Model204, y0, 0.0, days * 1440.0, q, d_sp)
        full_dense = full_dense.reshape(ns, len(q), 5)
        for sel in (None, (1, 4)):
            api.set_output_states(sel)
            width = 5 if sel is None else len(sel)
            out = np.full((ns, len(q), width), np.nan)

            def sink(qb, qe, rows):
                out[:, qb:qe] = rows
            final, info = api.run_rk45_streamed(H.Model204, y0, 0.0, days * 1440.0, q, d_sp, 50, sink, with_info=True)
            assert np.array_equal(final, full_final) and info["n_attempt"].sum() > 0
            assert np.array_equal(out, full_dense if sel is None else full_dense[:, :, list(sel)])
        api.set_radau_fallback(True)
        with pytest.raises(H.HlmError):
            api.run_rk45_streamed(H.Model204, y0, 0.0, days * 1440.0, q, d_sp, 50, None)
    finally:
        api.set_radau_fallback(False)
        api.set_output_states(None)
        f.free()
        d_sp.free()
    exe = os.path.join(ROOT, "tools", "rk45_solver")
    h5dump = os.path.join(os.environ.get("HDF5_DIR", "/opt/conda"), "bin", "h5dump")
    if not os.path.exists(exe) or not os.path.exists(h5dump):
        pytest.skip("driver or h5dump not available")
    base = [exe, "--model", "204", "--params", golden["small_test_csv"], "--tf", "2880", "--query-stride", "7", "--initial-step", "1e-6",
            "--out-format", "nc"]
    blobs = {}
    for tag, extra in (("whole", []), ("streamed", ["--stream-dense", "64"]), ("streamed_sel", ["--stream-dense", "64", "--states", "0,2"]),
                       ("whole_sel", ["--states", "0,2"])):
        d = tmp_path / tag
        d.mkdir()
        p = subprocess.run(base + extra + ["--out-final", str(d / "final.csv"), "--out-dense", str(d / "dense.csv")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        line = json.loads(p.stdout.strip().splitlines()[-1])
        assert line["streamed_dense_window"] == (64 if "streamed" in tag else 0)
        vals = []
        for name in ("final", "dense"):
            for ds in ("/outputs", "/variable", "/time") if name == "dense" else ("/outputs", "/variable"):
                b = str(d / f"{name}{ds.replace('/', '_')}.bin")
                subprocess.run([h5dump, "-d", ds, "-b", "LE", "-o", b, str(d / f"{name}.nc")], check=True, capture_output=True)
                vals.append(open(b, "rb").read())
        blobs[tag] = vals
    assert blobs["streamed"] == blobs["whole"] and blobs["streamed_sel"] == blobs["whole_sel"]


@pytest.mark.parametrize("layout", [0, 2])
def test_selection_on_the_three_state_model(gpu, layout):
    """The selecting kernels are templates over Model::N_EQ like everything else: ToyModel3 (N_EQ = 3, one forcing),
    every proper subset, all layouts and the Radau pass on its general (pivoted LU) route -- the selected columns of
    the full run, bit for bit."""
    H = gpu
    from test_parity_gpu import _toy_case
    from tiger_hlm_gpu_amd import synth
    from tiger_hlm_gpu_amd.engine import ShardEngine
    y0, sp, inflow, cells, tf = _toy_case()
    q = synth.minute_queries(0.0, tf, 17.0)

    def run(sel, radau=False):
        eng = ShardEngine(H.ToyModel3, y0, 0.0, tf, sp=sp, query=q, dense_tiled=layout, dense_states=sel)
        eng.set_forcings_per_cell((inflow,), cells, (1.0,), (inflow.shape[0],))
        eng.integrate(radau=radau)
        return eng.results()
    full = run(None)
    for sel in ((0,), (1,), (2,), (0, 1), (0, 2), (1, 2)):
        got = run(sel)
        assert got["dense"].shape == (len(y0), len(q), len(sel))
        assert np.array_equal(got["dense"], full["dense"][:, :, list(sel)]), (layout, sel)
        assert np.array_equal(got["y_final"], full["y_final"])
    with pytest.raises(H.HlmError):
        run((3,))


# ---- dense launches as a persistent grid over (sub-window, tile) items (hlm_device_job.dense_sub_windows) --------------
@pytest.mark.parametrize("case", ["wet-with-early-queries", "cold-cells-flagged", "ragged-sub-wave", "strict", "toy", "selected"])
def test_dense_sub_windows_hold_the_same_rows(gpu, case):
    """hlm_device_job.dense_sub_windows = n: the dense launch in the tile layout (csrc/hlm_rk45_dense_kernel.h) as a
    persistent grid over (sub-window, tile) items -- a lane pauses at the end of its sub-window exactly as at the end
    of a query window (steps that straddle it are given back and redone) and another wave continues the tile from the
    state record.  Rows, finals, flags and both counters equal the one-wave-per-tile launch bit for bit for n = 2, 3, 7
    and 64 (more sub-windows than some steps have queries): queries before t0 and past tf, cold cells with hillslopes
    flagged in different sub-windows (zero rows from there on), fewer hillslopes than a wave, STRICT arithmetic, the
    three-state model, a device-side output selection; and inside the windowed streaming to the host (the sub-windows
    then cut each window; the lanes live in the caller's record)."""
    H = gpu
    from tiger_hlm_gpu_amd import synth
    from tiger_hlm_gpu_amd.engine import ShardEngine
    kw, days, model = {}, 2, H.Model204
    if case == "toy":
        rng = np.random.default_rng(3)
        ns, days = 300, 3
        sp = synth.model204_params(0, ns, seed=3)
        sp["alpha3"] = rng.uniform(5.0, 60.0, ns)
        sp["alpha4"] = rng.uniform(20.0, 400.0, ns)
        y0 = np.stack([rng.uniform(0.0, 2.0, ns), rng.uniform(0.0, 1.0, ns), rng.uniform(0.0, 0.5, ns)], axis=1)
        cells = synth.cell_of_system(0, ns, ncell=16, run_len=7)
        inflow = ((rng.random((24 * days, 16)) < 0.3) * rng.uniform(0.0, 0.2, (24 * days, 16))).astype(np.float32)
        model = H.ToyModel3
        q = synth.minute_queries(0.0, days * 1440.0, 23.0)
    else:
        ns, tmin, wet = {"wet-with-early-queries": (300, 0.5, True), "cold-cells-flagged": (5000, -5.0, False), "ragged-sub-wave": (37, 0.5, True),
                         "strict": (700, 0.5, True), "selected": (1000, -5.0, True)}[case]
        days = 8 if case == "cold-cells-flagged" else 2
        sp, y0, cells, pr, t2m, dt, nT, q = _case(H, ns=ns, days=days, wet=wet, tmin=tmin)
        if case == "strict":
            kw = dict(math_mode=H.MATH_STRICT)
        if case == "selected":
            kw = dict(dense_states=(1, 3))

    def engine(n):
        eng = ShardEngine(model, y0, 0.0, days * 1440.0, sp=sp, query=q, dense_tiled=2, dense_sub_windows=n, **kw)
        if case == "toy":
            eng.set_forcings_per_cell((inflow,), cells, (1.0,), (inflow.shape[0],))
        else:
            eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)
        return eng
    e1 = engine(1)
    e1.integrate()
    one = e1.results()
    if case == "cold-cells-flagged":
        assert 0.02 < (one["stiff"] == 1).mean() < 0.98
    for n in (2, 3, 7, 64):
        e = engine(n)
        e.dense.fill_(-7.0)   # every row of the buffer is written by the launch
        e.integrate()
        got = e.results()
        for k in ("dense", "y_final", "stiff", "n_attempt", "n_accept"):
            assert np.array_equal(got[k], one[k]), (case, n, k)
    # windowed streaming to the host with sub-windows inside every window
    e = engine(3)
    host, _ = e.integrate_streamed(37)
    assert np.array_equal(host, one["dense"]) and np.array_equal(e.results()["y_final"], one["y_final"])


def test_automatic_dense_sub_windows_hold_the_same_rows_at_their_threshold_sizes(gpu):
    """hlm_device_job.dense_sub_windows = 0 (the default): the library cuts a dense launch in the tile layout into four
    sub-windows by itself when it has fewer than five rounds of waves and at least 256 queries (auto_dense_sub_windows,
    csrc/hlm_abi.hip) -- so the DEFAULT path of a config-5-sized run is the persistent (sub-window, tile) grid.  At
    sizes on both sides of the rule (round-4 advisor): a ragged last tile, cold cells (hillslopes flagged in different
    sub-windows), one launch and streamed query windows of 300 queries with the lanes in the caller's record (every
    window again cut in four) and of 200 (below the rule: one wave per tile) -- rows, finals, flags and counters equal
    the explicit one-wave-per-tile launch bit for bit."""
    H = gpu
    from tiger_hlm_gpu_amd.engine import ShardEngine
    ns, days = 10_037, 3
    sp, y0, cells, pr, t2m, dt, nT, q = _case(H, ns=ns, days=days, wet=False, tmin=-5.0)
    assert len(q) >= 256

    def engine(n):
        eng = ShardEngine(H.Model204, y0, 0.0, days * 1440.0, sp=sp, query=q, dense_tiled=2, dense_sub_windows=n)
        eng.expand_forcings_on_device((pr, t2m), cells, dt, nT)
        return eng
    e1 = engine(1)
    e1.integrate()
    one = e1.results()
    assert 0.02 < (one["stiff"] == 1).mean() < 0.98
    e0 = engine(0)
    e0.dense.fill_(-7.0)
    e0.integrate()
    got = e0.results()
    for k in ("dense", "y_final", "stiff", "n_attempt", "n_accept"):
        assert np.array_equal(got[k], one[k]), k
    # streamed windows in the TILE layout (stream path 2: the transposing transfer kernel writes pinned host memory), so
    # that every window launch is itself a candidate for the automatic rule; path 0 (linear windows) for comparison
    for path, W in ((2, 300), (2, 200), (0, 300)):
        H._lib.lib().hlm_set_stream_path(path)
        try:
            e = engine(0)
            host, _ = e.integrate_streamed(W)
        finally:
            H._lib.lib().hlm_set_stream_path(0)
        assert np.array_equal(host, one["dense"]), (path, W)
        r = e.results()
        for k in ("y_final", "stiff", "n_attempt", "n_accept"):
            assert np.array_equal(r[k], one[k]), (path, W, k)
