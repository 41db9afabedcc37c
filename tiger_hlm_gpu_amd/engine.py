"""Device-resident driver for one GPU's shard of hillslopes.

Holds the shard's inputs and outputs in HBM as torch tensors (torch = allocator and
stream plumbing only) and runs the HIP path through the C ABI
(hlm_rk45_integrate_device).  Used by bench.py, __graft_entry__.smoke() and the GPU
tests; raises if the HIP library or a device is missing -- there is no CPU fallback.
"""
import numpy as np
import torch

from . import _lib, api
from .models import Parameters, SP_DTYPE


class ShardEngine:
    def __init__(self, model, y0, t0, tf, sp=None, query=None, params=None, device="cuda:0",
                 math_mode=_lib.MATH_FAST, fixed_step=0.0, attempt_cap=0, alloc_dense=True,
                 dense_tiled=2, dense_states=None, time_pieces=None, queue_run_pieces=0, queue_group_tiles=0, dense_sub_windows=0, queue_hybrid=0):
        _lib.lib()  # fail loudly if the extension is not built
        if not torch.cuda.is_available():
            raise RuntimeError("ShardEngine needs a HIP device (torch.cuda.is_available() is False)")
        self.model = model
        self.device = torch.device(device)
        self.t0, self.tf = float(t0), float(tf)
        self.params = params if params is not None else Parameters()
        self.math_mode, self.fixed_step, self.attempt_cap = math_mode, fixed_step, attempt_cap
        # hlm_device_job.time_pieces (launches without dense output): a persistent grid over (time piece, tile) items
        # instead of one wave per tile; None = the host API's automatic rule (from 4 096 hillslopes on, one piece per
        # twelve simulated hours, at most 32, none below 8)
        self.time_pieces = time_pieces
        # hlm_device_job.queue_run_pieces: 0 = automatic runs of pieces per queue item, n = runs of n pieces (1 = every
        # piece handed over through HBM), -n = the automatic schedule with runs of at most n pieces
        self.queue_run_pieces = int(queue_run_pieces)
        # hlm_device_job.queue_group_tiles: G > 1 = an item is a group of G tiles whose survivors are packed per piece
        self.queue_group_tiles = int(queue_group_tiles)
        # hlm_device_job.dense_sub_windows: n > 1 = a dense launch (tile layout) as a persistent grid over (sub-window, tile) items
        self.dense_sub_windows = int(dense_sub_windows)
        # hlm_device_job.queue_hybrid: 1 = waves holding saturated lanes run one block of general stages, -1 = a block per form
        self.queue_hybrid = int(queue_hybrid)
        # hlm_device_job.d_path_counts (diagnostic): count_paths() allocates the 7 counters and the launches add to them
        self.path_counts = None
        y0 = np.ascontiguousarray(y0, dtype=np.float64).reshape(-1, model.N_EQ)
        self.ns = y0.shape[0]
        dev = self.device
        self.y0 = torch.from_numpy(y0).to(dev)
        self.sp = None
        if sp is not None:
            sp = np.ascontiguousarray(sp, dtype=SP_DTYPE)
            assert sp.shape == (self.ns,)
            self.sp = torch.from_numpy(sp.view(np.uint8).reshape(self.ns, SP_DTYPE.itemsize)).to(dev)
        self.query = None
        if query is not None and len(query) > 0:
            self.query = torch.from_numpy(np.ascontiguousarray(query, dtype=np.float64)).to(dev)
        self.nq = 0 if self.query is None else self.query.numel()
        self.forc = self.forc_col = None
        self.forc_dt, self.forc_nT, self.forc_ld = (), (), 0
        # y_final | stiff | n_attempt | n_accept live in ONE allocation (52 B per hillslope), so the
        # end-of-run gather to rank 0 is a single collective on `result_blob`
        ns8 = -(-self.ns // 2) * 2  # keeps every section 8-byte aligned
        self.result_blob = torch.empty(self.ns * 8 * model.N_EQ + 3 * ns8 * 4, dtype=torch.uint8, device=dev)
        o = self.ns * 8 * model.N_EQ
        self.y_final = self.result_blob[:o].view(torch.float64).view(self.ns, model.N_EQ)
        # device layout of the dense buffer (hlm_device_job.dense_tiled): 2 (default) tile-major
        # [tile][q][c][lane], the kernel with wave-synchronous stores; 0 host order, the free-running kernel.  results() / dense_host_order() deliver [ns][nq][5] whatever the layout.
        self.dense_tiled = int(dense_tiled)
        # output selection (hlm_device_job.dense_ncomp; the reference's output.states): the dense rows hold only
        # these state components -- K/N_EQ of the HBM and PCIe bytes; None = all
        self.dense_states = None if dense_states is None else tuple(int(c) for c in dense_states)
        self.dense_width = model.N_EQ if self.dense_states is None else len(self.dense_states)
        rows = -(-self.ns // 64) * 64 if self.dense_tiled else self.ns
        self.dense = (torch.empty((rows, self.nq, self.dense_width), dtype=torch.float64, device=dev)
                      if self.nq and alloc_dense else None)
        self.stiff = self.result_blob[o:o + ns8 * 4].view(torch.int32)[:self.ns]
        self.n_attempt = self.result_blob[o + ns8 * 4:o + 2 * ns8 * 4].view(torch.int32)[:self.ns]  # uint32 bits
        self.n_accept = self.result_blob[o + 2 * ns8 * 4:o + 3 * ns8 * 4].view(torch.int32)[:self.ns]

    # -- forcings ------------------------------------------------------------------------
    def set_forcings_per_stream(self, block, dt_hours, nT):
        """Reference layout float[sum nT][ns] (main.cpp:527-549); ``block`` may be a numpy
        array or a CUDA tensor already in HBM."""
        if not torch.is_tensor(block):
            block = torch.from_numpy(np.ascontiguousarray(block, dtype=np.float32))
        self.forc = block.to(self.device).contiguous().view(-1)
        assert self.forc.numel() == sum(nT) * self.ns
        self.forc_dt, self.forc_nT, self.forc_ld, self.forc_col = tuple(dt_hours), tuple(nT), self.ns, None

    def set_forcings_per_cell(self, tables, cells, dt_hours, nT):
        """Compact per-grid-cell tables float[sum nT][ncell] + column index per hillslope."""
        tab = np.concatenate([np.asarray(t, dtype=np.float32) for t in tables], axis=0)
        self.forc = torch.from_numpy(np.ascontiguousarray(tab)).to(self.device).view(-1)
        self.forc_ld = tab.shape[1]
        self.forc_col = torch.from_numpy(np.ascontiguousarray(cells, dtype=np.int32)).to(self.device)
        self.forc_dt, self.forc_nT = tuple(dt_hours), tuple(nT)

    def expand_forcings_on_device(self, tables, cells, dt_hours, nT):
        """Build the reference's per-stream layout in HBM from per-cell tables (what
        main.cpp:541-548 does on the host)."""
        cols = torch.from_numpy(np.ascontiguousarray(cells, dtype=np.int64)).to(self.device)
        parts = [torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32)).to(self.device).index_select(1, cols)
                 for t in tables]
        self.set_forcings_per_stream(torch.cat(parts, dim=0), dt_hours, nT)

    # -- run -------------------------------------------------------------------------------
    def job(self):
        j = api.make_device_job(
            self.model, y0=self.y0, t0=self.t0, tf=self.tf, query=self.query, sp=self.sp,
            forc=self.forc, forc_dt_hours=self.forc_dt, forc_nT=self.forc_nT, forc_ld=self.forc_ld,
            forc_col=self.forc_col, params=self.params, fixed_step=self.fixed_step,
            attempt_cap=self.attempt_cap, math_mode=self.math_mode, y_final=self.y_final,
            dense=self.dense, stiff=self.stiff, n_attempt=self.n_attempt, n_accept=self.n_accept,
            dense_states=self.dense_states)
        j.dense_tiled = self.dense_tiled if self.dense is not None else 0
        tp = self.time_pieces
        j.time_pieces = -1 if tp is None else int(tp)
        j.queue_run_pieces = self.queue_run_pieces
        j.queue_group_tiles = self.queue_group_tiles
        j.dense_sub_windows = self.dense_sub_windows
        j.queue_hybrid = self.queue_hybrid
        if self.dense is not None or self.nq:
            j.time_pieces = 0
        if self.path_counts is not None:
            j.d_path_counts = self.path_counts.data_ptr()
        return j

    def count_paths(self, enabled=True):
        """Route the launches through the counting instance of the queue kernel (hlm_device_job.d_path_counts: FAST,
        adaptive, time pieces, no dense output) and zero the counters; ``path_shares()`` reads them."""
        self.path_counts = torch.zeros(_lib.PATH_COUNTERS, dtype=torch.int64, device=self.device) if enabled else None

    def path_shares(self):
        """dict of the counters as shares of the lane-attempts / wave-attempts they count (see include/hlm_hip.h)."""
        torch.cuda.synchronize(self.device)
        n = [int(v) for v in self.path_counts.cpu().numpy()]
        lanes, waves = max(sum(n[:4]), 1), max(n[5], 1)
        return {"lane_attempts": sum(n[:4]), "unsaturated_dry_surface_closed_form": n[0] / lanes, "unsaturated_wet_surface_closed_form": n[1] / lanes,
                "dry_surface_stagewise": n[2] / lanes, "general_rhs": n[3] / lanes, "computed_twice": n[4] / lanes,
                "wave_attempts": n[5], "wave_attempts_on_several_forms": n[6] / waves,
                "mean_lanes_per_wave_attempt": sum(n[:4]) / waves}

    def integrate(self, stream=None, t_stop=None, resume_in=False, radau=False):
        """Enqueue one pass of the hot path on ``stream`` (default: torch's current stream).
        ``t_stop``: pause every hillslope once its t reaches it (state kept in the engine's
        resume buffer); ``resume_in``: continue from that buffer.  A sequence of such launches
        gives the same bits as one uninterrupted pass.  ``radau``: follow the pass with the
        Radau-IIA fallback over the hillslopes it flagged as stiff -- True: each restarts at t0 (the
        reference's kernel, radau_kernel.cu:44-52); "continue": each continues from the state it had reached
        (hlm_device_job.d_flag_state), the hillslopes flagged earliest first."""
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        j = self.job()
        if radau == "continue":
            if getattr(self, "_flag_state", None) is None:
                self._flag_state = torch.zeros(self.ns * (self.model.N_EQ + 1), dtype=torch.float64, device=self.device)
            j.d_flag_state = self._flag_state.data_ptr()
        if t_stop is not None or resume_in:
            j.d_resume, j.resume_in = self._resume_buffer().data_ptr(), int(bool(resume_in))
            if t_stop is not None:
                j.use_t_stop, j.t_stop = 1, float(t_stop)
        api.integrate_device(j, s)
        if radau:
            api.radau_fallback_device(j, s)

    def _resume_buffer(self):
        if getattr(self, "_resume", None) is None:
            self._resume = torch.zeros(api.resume_bytes(self.ns, self.model), dtype=torch.uint8, device=self.device)
        return self._resume

    def checkpoint(self):
        """Host copy of the integrator state of a paused run (76 bytes per hillslope)."""
        torch.cuda.synchronize(self.device)
        return self._resume_buffer().cpu().numpy().copy()

    def restore(self, blob):
        self._resume_buffer().copy_(torch.from_numpy(np.ascontiguousarray(blob, dtype=np.uint8)))

    def integrate_forcing_streamed_abi(self, tables, dt_hours, chunk_minutes, cells=None):
        """Same as integrate_forcing_streamed, but the chunk loop runs inside the library
        (hlm_rk45_integrate_forcing_streamed): what a C / C++ host calls."""
        import ctypes as C
        host = [self.pin_forcing_table(t) for t in tables]
        ld = int(host[0].shape[1])
        col = None if cells is None else torch.from_numpy(np.ascontiguousarray(cells, dtype=np.int32)).to(self.device)
        j = self.job()
        j.d_forc, j.d_forc_col, j.forc_ld, j.nforc = None, (None if col is None else col.data_ptr()), ld, len(host)
        for k, h in enumerate(host):
            j.forc_dt_hours[k], j.forc_nT[k] = float(dt_hours[k]), int(h.shape[0])
        ptrs = (C.c_void_p * len(host))(*[h.data_ptr() for h in host])
        n = C.c_int(0)
        s = torch.cuda.current_stream(self.device)
        _lib.check(_lib.lib().hlm_rk45_integrate_forcing_streamed(C.byref(j), ptrs, float(chunk_minutes), s.cuda_stream, C.byref(n)))
        return n.value

    @staticmethod
    def pin_forcing_table(t):
        """float32 [nT][ld] table in pinned host memory (a tensor that already is one is used as is,
        so a caller streaming the same series repeatedly pins it once)."""
        if torch.is_tensor(t) and t.is_pinned() and t.dtype == torch.float32 and t.is_contiguous():
            return t
        if torch.is_tensor(t):
            t = t.cpu().numpy()
        return torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32)).pin_memory()

    def integrate_forcing_streamed(self, tables, dt_hours, chunk_minutes, cells=None):
        """BASELINE config 3, "forcings streamed": the forcing series stay in (pinned) host
        memory and pass through HBM in time chunks.  While chunk k integrates (lanes pause at
        its end time), chunk k+1 is uploaded on a second stream into the other of two device
        buffers.  ``tables``: per forcing a float32 array [nT_j][ld]; ``cells``: int32 column
        index per hillslope when the tables are per grid cell (ld = ncell), None when they are
        per stream (ld = ns, the reference layout).  Bit-identical to the resident-forcing run."""
        dev = self.device
        host = [self.pin_forcing_table(t) for t in tables]
        nT = [int(h.shape[0]) for h in host]
        ld = int(host[0].shape[1])
        assert all(h.shape[1] == ld for h in host) and (cells is not None or ld == self.ns)
        col = None if cells is None else torch.from_numpy(np.ascontiguousarray(cells, dtype=np.int32)).to(dev)
        dtmin = [float(d) * 60.0 for d in dt_hours]

        def span(j, ta, tb):  # sample rows any t in [ta, tb] can ask for (rk45_kernel.cu:90-98)
            first = min(max(int(np.floor(ta / dtmin[j])), 0), nT[j] - 1)
            last = min(max(int(np.floor(tb / dtmin[j])), 0), nT[j] - 1)
            return first, last - first + 1

        # edges as t0 + k * step (never by repeated addition), piece count bounded up front
        npieces = int(np.ceil((self.tf - self.t0) / float(chunk_minutes))) if self.tf > self.t0 else 0
        if npieces > (1 << 20):
            raise ValueError(f"chunk_minutes = {chunk_minutes} cuts the window into {npieces} pieces (at most 2^20)")
        edges = [self.t0] + [e for e in (self.t0 + k * float(chunk_minutes) for k in range(1, npieces)) if self.t0 < e < self.tf]
        if npieces > 0:
            edges.append(self.tf)
        max_rows = sum(max(span(j, a, b)[1] for a, b in zip(edges[:-1], edges[1:])) for j in range(len(host)))
        bufs = [torch.empty(max_rows * ld, dtype=torch.float32, device=dev) for _ in range(2)]
        compute = torch.cuda.current_stream(dev)
        copy = torch.cuda.Stream(device=dev)
        done = [None, None]
        for k, (ta, tb) in enumerate(zip(edges[:-1], edges[1:])):
            b = k & 1
            spans = [span(j, ta, tb) for j in range(len(host))]
            with torch.cuda.stream(copy):
                if done[b] is not None:
                    copy.wait_event(done[b])  # the kernel that read this buffer has finished
                off = 0
                for j, (first, rows) in enumerate(spans):
                    bufs[b][off:off + rows * ld].copy_(host[j][first:first + rows].reshape(-1), non_blocking=True)
                    off += rows * ld
                uploaded = torch.cuda.Event()
                uploaded.record(copy)
            compute.wait_event(uploaded)
            j_ = self.job()
            j_.d_forc, j_.d_forc_col, j_.forc_ld, j_.nforc = bufs[b].data_ptr(), (None if col is None else col.data_ptr()), ld, len(host)
            for j, (first, rows) in enumerate(spans):
                j_.forc_dt_hours[j], j_.forc_nT[j] = float(dt_hours[j]), nT[j]
                j_.forc_first_index[j], j_.forc_rows[j] = first, rows
            j_.d_resume, j_.resume_in = self._resume_buffer().data_ptr(), int(k > 0)
            if tb < self.tf:
                j_.use_t_stop, j_.t_stop = 1, tb
            api.integrate_device(j_, compute)
            done[b] = torch.cuda.Event()
            done[b].record(compute)
        torch.cuda.synchronize(dev)
        return len(edges) - 1

    def integrate_streamed(self, window_queries, host_out=None, stream=None):
        """BASELINE config 5: dense output streamed to pinned host memory in windows of
        ``window_queries`` queries, D2H of window k overlapped with the integration of
        window k+1.  Returns (host array [ns, nq, 5], stats dict).  The engine's own
        ``dense`` buffer is not used (it may not even fit in HBM)."""
        assert self.nq > 0
        if host_out is None:
            host_out = torch.empty((self.ns, self.nq, self.dense_width), dtype=torch.float64, pin_memory=True)
        arr = host_out.numpy() if torch.is_tensor(host_out) else host_out
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        keep, self.dense = self.dense, None
        try:
            job = self.job()
        finally:
            self.dense = keep
        stats = api.integrate_streamed(job, window_queries, arr, s)
        return arr, stats

    def integrate_streamed_sink(self, window_queries, sink=None, stream=None):
        """BASELINE config 5 at sizes that fit no host array: the dense rows pass window by window through the
        library's pinned staging buffers into ``sink(query_begin, query_end, rows[ns, W, K])`` (None = discard)."""
        assert self.nq > 0
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        keep, self.dense = self.dense, None
        try:
            job = self.job()
        finally:
            self.dense = keep
        return api.integrate_streamed_sink(job, window_queries, sink, s)

    def total_attempts(self):
        return int(self.n_attempt.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())

    @staticmethod
    def split_result_blob(blob, ns, n_eq=5):
        """Views (y_final, stiff, n_attempt, n_accept) of a result_blob of ``ns`` hillslopes."""
        ns8 = -(-ns // 2) * 2
        o = ns * 8 * n_eq
        y = blob[:o].view(torch.float64).view(ns, n_eq)
        i32 = lambda k: blob[o + k * ns8 * 4:o + (k + 1) * ns8 * 4].view(torch.int32)[:ns]
        return y, i32(0), i32(1), i32(2)

    def dense_host_order(self):
        """The dense buffer as a [ns, nq, 5] device tensor whatever its device layout."""
        if self.dense is None or self.dense_tiled == 0:
            return self.dense
        tiles = self.dense.shape[0] // 64
        n_eq = self.dense_width
        if self.dense_tiled == 1:
            d = self.dense.view(tiles, self.nq, 64, n_eq).permute(0, 2, 1, 3)  # [tile][q][lane][c]
        else:
            d = self.dense.view(tiles, self.nq, n_eq, 64).permute(0, 3, 1, 2)  # [tile][q][c][lane]
        return d.reshape(tiles * 64, self.nq, n_eq)[:self.ns]

    def results(self):
        """Host copies: dict(y_final[ns,5], dense[ns,nq,5] or None, stiff, n_attempt, n_accept)."""
        torch.cuda.synchronize(self.device)
        u32 = lambda t: t.cpu().numpy().view(np.uint32)
        dense = self.dense_host_order()
        return {"y_final": self.y_final.cpu().numpy(),
                "dense": None if dense is None else dense.cpu().numpy(),
                "stiff": self.stiff.cpu().numpy(), "n_attempt": u32(self.n_attempt),
                "n_accept": u32(self.n_accept)}
