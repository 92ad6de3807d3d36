"""Python host of the hypothesis engine (``caf_plan_*`` in include/caf.h).

``CAFPlan`` is the stateful create -> run many -> destroy object behind the
reference-signature entry points in ``xcorrRoutines.py`` (GroupXcorr,
GroupXcorrFFT, cztXcorr, TemplateCrossCorrelator, fastXcorr's restricted-bin
mode) and behind ``bench.py``.  It owns no algorithm: every number comes out of
libcaf.so.
"""

import ctypes as ct

import numpy as np

from . import _lib
from .devarray import DeviceArray, as_device_ptr, asarray, empty, zeros


class CAFResult:
    """Device-resident outputs of one execute."""

    __slots__ = ("surface", "surface_t", "row_max", "row_arg", "peak_val", "peak_delay", "peak_freq", "cqf", "_row_arg_zeroed")

    def __init__(self):
        self.surface = self.surface_t = self.row_max = self.row_arg = None
        self._row_arg_zeroed = None  # the F == 1 argument array that run() allocated itself and has already filled with zeros
        self.peak_val = self.peak_delay = self.peak_freq = None
        self.cqf = None


class CAFPlan:
    """QF^2(t, s, f) for T templates x F frequency hypotheses over a range of delays.

    Parameters
    ----------
    templates : complex64 (T, N) or (N,)  -- NOT conjugated unless autoConj=False
    bins, grid : on-grid hypotheses nu_f = bins[f]/grid (every bins[f] * block / grid must be whole), or
    freqs_norm : arbitrary hypotheses in cycles/sample (f / fs)
    group_starts, group_lens : support of a composite template (GroupXcorr semantics)
    max_rx_len : largest rx length that will be passed to run()
    """

    def __init__(
        self,
        templates,
        max_rx_len,
        bins=None,
        grid=None,
        freqs_norm=None,
        group_starts=None,
        group_lens=None,
        autoConj=True,
        log2_block=0,
        blocks_per_batch=0,
        device=None,
        engine="auto",
    ):
        lib = _lib.load()
        _lib.require_device()
        if device is not None:
            _lib.check(lib.caf_set_device(int(device)), "caf_set_device")
        tm = np.ascontiguousarray(np.atleast_2d(templates), dtype=np.complex64)
        if tm.ndim != 2:
            raise ValueError("templates must be 1-D or 2-D")
        self.T, self.N = tm.shape
        d = _lib.CafPlanDesc()
        d.num_templates, d.template_len = self.T, self.N
        d.h_templates = tm.ctypes.data
        d.auto_conj = 1 if autoConj else 0
        keep = [tm]
        if group_starts is not None:
            gs = np.ascontiguousarray(group_starts, dtype=np.int32)
            gl = np.ascontiguousarray(group_lens, dtype=np.int32)
            if gs.shape != gl.shape or gs.ndim != 1:
                raise ValueError("group_starts and group_lens must be 1-D of equal length")
            d.num_groups = gs.size
            d.h_group_start, d.h_group_len = gs.ctypes.data, gl.ctypes.data
            keep += [gs, gl]
        else:
            d.num_groups = 0
        if (bins is None) == (freqs_norm is None):
            raise ValueError("give exactly one of bins= or freqs_norm=")
        if bins is not None:
            b = np.ascontiguousarray(bins, dtype=np.int32)
            d.freq_mode, d.num_freqs = _lib.CAF_FREQ_BINS, b.size
            d.h_bins, d.grid = b.ctypes.data, int(grid if grid is not None else self.N)
            keep.append(b)
        else:
            fr = np.ascontiguousarray(freqs_norm, dtype=np.float64)
            d.freq_mode, d.num_freqs = _lib.CAF_FREQ_NORM, fr.size
            d.h_freqs_norm = fr.ctypes.data
            keep.append(fr)
        self.F = int(d.num_freqs)
        d.max_rx_len = int(max_rx_len)
        d.log2_block, d.blocks_per_batch = int(log2_block), int(blocks_per_batch)
        try:
            d.engine = _lib.ENGINE_IDS[engine]
        except KeyError:
            raise ValueError("engine must be one of %s" % sorted(_lib.ENGINE_IDS))
        self.engine = engine
        h = ct.c_void_p()
        _lib.check(lib.caf_plan_create(ct.byref(h), ct.byref(d)), "caf_plan_create")
        self._h = h
        self.max_rx_len = int(max_rx_len)
        blk, step, nb, ws = ct.c_int32(), ct.c_int32(), ct.c_int32(), ct.c_int64()
        _lib.check(lib.caf_plan_info(h, ct.byref(blk), ct.byref(step), ct.byref(nb), ct.byref(ws)))
        self.block, self.step, self.blocks_per_batch, self.workspace_bytes = blk.value, step.value, nb.value, ws.value
        eng = ct.c_int32()
        _lib.check(lib.caf_plan_engine(h, ct.byref(eng)))
        self.engine_used = _lib.ENGINE_NAMES[eng.value]

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.load().caf_plan_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------
    def run(self, rx, shift_start=0, num_shifts=None, surface=False, rows=True, peak=True, stream=None, out=None,
            cqf=False, surface_t=False):
        """Asynchronous execute on device-resident rx (DeviceArray or CUDA torch tensor).

        Returns a CAFResult of DeviceArrays: surface (T,S,F) float32 if requested, row_max (T,S)
        float32 / row_arg (T,S) int32 if ``rows`` (``rows="max"``: row_max only), peak_val/peak_delay/peak_freq (T,)
        if ``peak``.
        Pass ``out`` (a previous CAFResult) to reuse its buffers.
        """
        ptr, nbytes = as_device_ptr(rx)
        rx_len = nbytes // 8
        if num_shifts is None:
            num_shifts = rx_len - self.N + 1 - shift_start
        S = int(num_shifts)
        res = out if out is not None else CAFResult()
        if surface and res.surface is None:
            res.surface = empty((self.T, S, self.F), np.float32)
        if surface_t and res.surface_t is None:
            res.surface_t = empty((self.T, self.F, S), np.float32)
        want_arg = bool(rows) and rows != "max"  # rows="max": the per-delay maxima only (no argument array)
        if rows and res.row_max is None:
            res.row_max = empty((self.T, S), np.float32)
        own_arg = False
        if want_arg and res.row_arg is None:
            res.row_arg = empty((self.T, S), np.int32)
            own_arg = True
        skip_arg = False
        if want_arg and self.F == 1:
            # one hypothesis per template: the argument of every per-delay maximum is 0.  An array that THIS method allocated
            # is zeroed once (the fill completed before the call returns, so that runs on any stream may follow) and not
            # handed to the library again (which would fill T x S x 4 bytes on every call: 0.65 ms of C3's 4.5).  An array the
            # caller put into the result object -- it may have been written since, or be a new tensor at an old address --
            # always goes to the library.
            if own_arg:
                lib = _lib.load()
                sp = ct.c_void_p(stream) if stream else None
                _lib.check(lib.caf_memset(ct.c_void_p(res.row_arg.ptr), 0, res.row_arg.nbytes, sp), "caf_memset")
                _lib.check(lib.caf_stream_sync(sp), "caf_stream_sync")
                res._row_arg_zeroed = res.row_arg
            skip_arg = res._row_arg_zeroed is res.row_arg
        elif want_arg:
            res._row_arg_zeroed = None  # (a plan with F > 1 writes real arguments into the shared result object)
        if peak and res.peak_val is None:
            res.peak_val = empty((self.T,), np.float32)
            res.peak_delay = empty((self.T,), np.int32)
            res.peak_freq = empty((self.T,), np.int32)
        if cqf and res.cqf is None:
            res.cqf = empty((self.T, self.F, S), np.complex64)
        o2 = _lib.CafOutputs2()
        o = o2.base
        o.d_cqf = res.cqf.ptr if cqf else None
        o.d_surface = res.surface.ptr if surface else None
        o.d_row_max = res.row_max.ptr if rows else None
        o.d_row_arg = res.row_arg.ptr if want_arg and not skip_arg else None
        o2.d_surface_t = res.surface_t.ptr if surface_t else None
        o.d_peak_val = res.peak_val.ptr if peak else None
        o.d_peak_delay = res.peak_delay.ptr if peak else None
        o.d_peak_freq = res.peak_freq.ptr if peak else None
        _lib.check(
            _lib.load().caf_plan_execute2(self._h, ct.c_void_p(ptr), rx_len, int(shift_start), S, ct.byref(o2),
                                         ct.c_void_p(stream) if stream else None),
            "caf_plan_execute",
        )
        return res

    def run_host(self, rx, shift_start=0, num_shifts=None, surface=False, rows=True, peak=True):
        """Blocking host-array call in the reference's DLL style: NumPy in, caller-allocated NumPy out."""
        rx = np.ascontiguousarray(rx, dtype=np.complex64)
        if num_shifts is None:
            num_shifts = rx.size - self.N + 1 - shift_start
        S = int(num_shifts)
        if S < 1:
            raise ValueError("no delays to evaluate")
        T, F = self.T, self.F
        surf = np.empty((T, S, F), np.float32) if surface else None
        rmax = np.empty((T, S), np.float32) if rows else None
        rarg = np.empty((T, S), np.int32) if rows else None
        pv = np.empty(T, np.float32) if peak else None
        pd = np.empty(T, np.int32) if peak else None
        pf = np.empty(T, np.int32) if peak else None
        p = lambda a: a.ctypes.data if a is not None else None  # noqa: E731
        _lib.check(
            _lib.load().caf_plan_execute_host(self._h, rx.ctypes.data, rx.size, int(shift_start), S, p(surf), p(rmax),
                                              p(rarg), p(pv), p(pd), p(pf)),
            "caf_plan_execute_host",
        )
        return {"surface": surf, "row_max": rmax, "row_arg": rarg, "peak_val": pv, "peak_delay": pd, "peak_freq": pf}

    # ------------------------------------------------------------------------------------
    def watchdog(self):
        """(mark, fft_next) of the persistent engine's polling watchdog: (0, 0) unless its hand-off protocol broke."""
        m = (ct.c_int32 * 2)()
        _lib.check(_lib.load().caf_plan_watchdog(self._h, m), "caf_plan_watchdog")
        return int(m[0]), int(m[1])

    def profile(self, enable=True):
        _lib.check(_lib.load().caf_plan_profile(self._h, 1 if enable else 0))

    def profile_get(self):
        ms = (ct.c_double * _lib.CAF_NUM_STAGES)()
        n = (ct.c_int64 * _lib.CAF_NUM_STAGES)()
        _lib.check(_lib.load().caf_plan_profile_get(self._h, ms, n))
        return {name: (ms[i], n[i]) for i, name in enumerate(_lib.STAGE_NAMES)}


__all__ = ["CAFPlan", "CAFResult", "DeviceArray", "asarray"]
