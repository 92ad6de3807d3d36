"""Drop-in host layer: the reference's xcorr call signatures (xcorrRoutines.py,
cython_ext/CyIppXcorrFFT, cython_ext/CyGroupXcorrFFT, pybinds/ippGroupXcorrCZT) in front of
libcaf.so.  Same names, argument meaning, return dtypes / shapes and error behaviour; every
number is computed on the MI355X (no NumPy fallback -- a missing library or GPU raises).

Two device strategies sit behind these entry points:
  * hypothesis engine (``CAFPlan``): overlap-save FFT correlation per (template, frequency)
    hypothesis -- used when the frequency set is restricted (GroupXcorr*, cztXcorr,
    TemplateCrossCorrelator, fastXcorr without frequency search);
  * per-delay path (``caf_xcorr_perdelay``): sliding product -> N-point row FFT -> |.|^2 / argmax
    -- the reference's literal algorithm, used when every FFT bin of every delay is wanted
    (fastXcorr freqsearch branches, cp_fastXcorr*, CyIppXcorrFFT).

GPU arithmetic is complex64/float32 (energies float64); float64/complex128 outputs of the CPU
signatures are widened copies of those results.
"""

import ctypes as ct
import os

import numpy as np

from . import _lib
from .caf import CAFPlan
from .cupyExtensions import (  # noqa: F401  (re-exported like `from cupyExtensions import *` upstream)
    cupyArgmaxAbsRows_complex64,
    cupyComplexMagnSq,
    cupyCopyGroups32fc,
    cupyCopyIncrementalEqualSlicesToMatrix_32fc,
    fftRows,
    multiplySlicesOptimistically,
    multiplySlidesNormalised,
    multiTemplateSlidingDotProduct,
)
from .devarray import DeviceArray, asarray, empty, requireDeviceArray, requireDtype, zeros
from .filterRoutines import cupyMovingAverage  # noqa: F401
from .signalCreationRoutines import makeFreq
from .spectralRoutines import CZTCached, CZTCachedGPU, next_fast_len  # noqa: F401

_MAX_PLANE_ELEMS = 1 << 27  # bound on (delays x bins) elements materialised per device call


# ------------------------------------------------------------------------------------------
# small helpers
# ------------------------------------------------------------------------------------------
def _c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


def _runs(shifts):
    """Split an index array into maximal arithmetic-progression runs: [(offset, start, step, count)]."""
    s = np.asarray(shifts, dtype=np.int64).reshape(-1)
    n = s.size
    if n == 0:
        return []
    if n == 1:
        return [(0, int(s[0]), 1, 1)]
    d = np.diff(s)
    if d[0] != 0 and np.all(d == d[0]):  # one progression (the usual call: a million delays cost a Python loop 125 ms)
        return [(0, int(s[0]), int(d[0]), n)]
    out, i = [], 0
    while i < n:
        if i + 1 >= n:
            out.append((i, int(s[i]), 1, 1))
            break
        step = int(d[i])
        if step == 0:
            out.append((i, int(s[i]), 1, 1))
            i += 1
            continue
        brk = np.flatnonzero(d[i:] != step)  # the run ends in front of the first other difference
        j = i + (int(brk[0]) if brk.size else n - 1 - i)
        out.append((i, int(s[i]), step, j - i + 1))
        i = j + 1
    return out


def _host_take(d_row, rel, dtype):
    """``d_row[rel]`` (a 1-D device array) as a host array of ``dtype``: a contiguous run in order is downloaded straight into the
    result -- float32 widened to float64 by the transfer itself -- instead of download + fancy-index copy + ``astype``."""
    rel = np.asarray(rel)
    n = rel.size
    if n and int(rel[-1]) - int(rel[0]) + 1 == n and (n == 1 or np.all(np.diff(rel) == 1)):
        lo = int(rel[0])
        if d_row.dtype == np.float32 and np.dtype(dtype) == np.float64:
            out = np.empty(n, np.float64)
            _lib.check(_lib.load().caf_d2h_f64(out.ctypes.data, ct.c_void_p(d_row.ptr + 4 * lo), n, None), "caf_d2h_f64")
            return out
        return d_row[lo : lo + n].get().astype(dtype, copy=False)
    return d_row.get()[rel].astype(dtype, copy=False)


def _perdelay(d_cut, n, d_rx, rx_len, start, step, num, zero_oor, want_qf2, want_idx, want_caf, want_ccaf,
              device_out=False):
    """One caf_xcorr_perdelay call; returns host arrays (DeviceArrays with ``device_out``)."""
    lib = _lib.load()
    qf2 = empty(num, np.float32) if want_qf2 else None
    idx = empty(num, np.int32) if want_idx else None
    caf = empty((num, n), np.float32) if want_caf else None
    ccaf = empty((num, n), np.complex64) if want_ccaf else None
    p = lambda a: ct.c_void_p(a.ptr) if a is not None else None  # noqa: E731
    _lib.check(
        lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), rx_len, int(start), int(step), int(num), 1 if zero_oor else 0,
                               p(qf2), p(idx), p(caf), p(ccaf), 0, None),
        "caf_xcorr_perdelay",
    )
    if device_out:
        return qf2, idx, caf, ccaf
    g = lambda a: a.get() if a is not None else None  # noqa: E731
    return g(qf2), g(idx), g(caf), g(ccaf)


def _engine_range(shifts):
    s = np.asarray(shifts, dtype=np.int64)
    lo, hi = int(s.min()), int(s.max())
    return lo, hi - lo + 1, s - lo


def _dev_index(rel, n):
    """None when ``rel`` selects 0..n-1 in order (no gather needed), else the int32 index array on the device."""
    rel = np.asarray(rel)
    if rel.size == n and (n == 0 or (rel[0] == 0 and rel[-1] == n - 1 and np.all(np.diff(rel) == 1))):
        return None
    return asarray(rel.astype(np.int32))


def _take_f64(d_f32, rel, out=None):
    """float64 device array of d_f32[rel] (d_f32: 1-D float32 DeviceArray), on the device."""
    rel = np.asarray(rel)
    out = empty(rel.size, np.float64) if out is None else out
    d_idx = _dev_index(rel, d_f32.size)
    _lib.check(_lib.load().caf_gather_f32_f64(ct.c_void_p(d_f32.ptr), d_f32.size, ct.c_void_p(d_idx.ptr) if d_idx is not None
                                              else None, rel.size, ct.c_void_p(out.ptr), None), "caf_gather_f32_f64")
    return out


def _take_u32(d_i32, rel, out=None):
    """uint32 device array of d_i32[rel] (non-negative int32 values: same bits), on the device."""
    rel = np.asarray(rel)
    out = empty(rel.size, np.uint32) if out is None else out
    d_idx = _dev_index(rel, d_i32.size)
    lib = _lib.load()
    if d_idx is None:
        _lib.check(lib.caf_d2d(ct.c_void_p(out.ptr), ct.c_void_p(d_i32.ptr), 4 * rel.size, None), "caf_d2d")
    else:
        _lib.check(lib.caf_gather_b32(ct.c_void_p(d_i32.ptr), d_i32.size, ct.c_void_p(d_idx.ptr), rel.size,
                                      ct.c_void_p(out.ptr), None), "caf_gather_b32")
    return out


def _host_surface(plan, d_rx, lo, cnt, rel, dtype):
    """Rows ``rel`` of the (cnt, F) QF^2 surface of a one-template plan over the delays lo .. lo + cnt - 1, as a host array
    of ``dtype`` (float32 / float64) -- what the reference's CPU signatures return (xcorrRoutines.py:553-566, 1028-1039).

    Where the engine can write the hypothesis-major surface itself (persistent engine, 16384-point blocks: no |y|^2 tiles,
    no tile role -- 10.7 instead of 13.4 ms at config C2) that launch is used, and the transposition to the reference's
    (delays, frequencies) layout happens on the host side of the download (``caf_d2h_transposed``: DMA lanes into pinned
    staging, 8 x 8 register transposes straight into the result array, float64 widening included).  Same numbers as the
    delay-major launch (tests/test_gpu_fullsize.py::test_c2_hypothesis_major_surface)."""
    rel = np.asarray(rel)
    F = plan.F
    lib = _lib.load()
    # (CAF_HOST_SURFACE_DELAY_MAJOR=1: the delay-major launch + plain download + NumPy selection / widening of rounds 1-4 --
    #  an A/B switch for scripts/time_host_surface.py)
    if plan.T == 1 and F > 1 and plan.engine_used == "persistent" and plan.block == 16384 \
            and os.environ.get("CAF_HOST_SURFACE_DELAY_MAJOR") != "1":
        res = plan.run(d_rx, shift_start=lo, num_shifts=cnt, surface_t=True, rows=False, peak=False)
        contiguous = rel.size > 0 and rel[-1] - rel[0] + 1 == rel.size and (rel.size == 1 or np.all(np.diff(rel) == 1))
        c0, nc = (int(rel[0]), int(rel.size)) if contiguous else (0, cnt)
        out = np.empty((nc, F), dtype)
        if nc:
            _lib.check(lib.caf_d2h_transposed(out.ctypes.data, 1 if out.dtype == np.float64 else 0, ct.c_void_p(res.surface_t.ptr),
                                              F, cnt, c0, nc, None), "caf_d2h_transposed")
        return out if contiguous else out[rel]
    # (the other engines: the delay-major surface is the reference's layout already; a contiguous run of delays is downloaded
    #  straight into the result -- widened to float64 by the transfer lanes, caf_d2h_f64 -- instead of download + selection +
    #  astype: three passes over what can be 17 GB)
    res = plan.run(d_rx, shift_start=lo, num_shifts=cnt, surface=True, rows=False, peak=False)
    contiguous = rel.size > 0 and rel[-1] - rel[0] + 1 == rel.size and (rel.size == 1 or np.all(np.diff(rel) == 1))
    if plan.T == 1 and contiguous and np.dtype(dtype) in (np.dtype(np.float32), np.dtype(np.float64)) \
            and os.environ.get("CAF_HOST_SURFACE_DELAY_MAJOR") != "1":
        out = np.empty((int(rel.size), F), dtype)
        src = ct.c_void_p(res.surface.ptr + int(rel[0]) * F * 4)
        if out.dtype == np.float64:
            _lib.check(lib.caf_d2h_f64(out.ctypes.data, src, out.size, None), "caf_d2h_f64")
        else:
            _lib.check(lib.caf_d2h(out.ctypes.data, src, out.size * 4, None), "caf_d2h")
        return out
    return res.surface.get()[0][rel].astype(dtype, copy=False)


# ------------------------------------------------------------------------------------------
# fastXcorr and relatives (host-array signatures)
# ------------------------------------------------------------------------------------------
def _complex_qf_plan(templates, max_rx_len, grid):
    """The plan behind the complex-QF plane (`caf_outputs.d_cqf`: fastXcorr(absResult=False), TemplateCrossCorrelator.correlate):
    the one-launch in-LDS engine writes it from its own work items -- 16384-point blocks up to 8192 samples, the chained roles
    (32768- / 65536-point blocks, partitions) up to 262144 --, the rocFFT engine otherwise (and under the A/B switches that
    take AUTO plans off the persistent engine).  The engine actually chosen is checked."""
    n = templates.shape[-1]
    if n <= 262144:
        plan = CAFPlan(templates, max_rx_len=max_rx_len, bins=[0], grid=grid)
        if plan.engine_used == "persistent":
            return plan
        plan.close()
    return CAFPlan(templates, max_rx_len=max_rx_len, bins=[0], grid=grid, engine="rocfft")


def fastXcorr(cutout, rx, freqsearch=False, outputCAF=False, shifts=None, absResult=True):
    """ref: xcorrRoutines.py:460-580 -- all six branches, same return dtypes:
    A float64[S]; A' complex128[S]; B (float64[S], uint32[S]); B' (complex128[S], uint32[S]);
    C float64[S, N]; C' complex128[S, N]."""
    cutout = np.asarray(cutout)
    rx = np.asarray(rx)
    n = len(cutout)
    all_delays = shifts is None
    if all_delays and not freqsearch:
        # every delay, no frequency search (the reference's default call): no index array is built for a million delays -- the
        # engine's range is the whole run and the result is downloaded straight into the array that is returned
        ns = max(len(rx) - n + 1, 0)
        shifts = None
    else:
        if all_delays:
            shifts = np.arange(len(rx) - n + 1)
        shifts = np.asarray(shifts)
        ns = len(shifts)
        if ns and (shifts.min() < 0 or shifts.max() + n > len(rx)):
            raise ValueError("shifts must keep the cutout inside rx")
    d_rx = asarray(_c64(rx))

    if not freqsearch:
        out = np.empty(ns, dtype=np.float64 if absResult else np.complex128)
        if ns == 0:
            return out
        if shifts is None:
            lo, cnt, rel = 0, ns, None
        else:
            lo, cnt, rel = _engine_range(shifts)
            if cnt == ns and (ns == 1 or np.all(np.diff(rel) == 1)):
                rel = None  # (a contiguous run in order)
        # (the complex QF plane comes from the one-launch in-LDS engine's own work items -- fused_item MODE 4, the chained roles
        #  beyond 8192 samples -- and from the rocFFT engine beyond 262144)
        grid = 1 << int(np.ceil(np.log2(max(n, 2))))
        plan = CAFPlan(_c64(cutout), max_rx_len=len(rx), bins=[0], grid=grid) if absResult else _complex_qf_plan(_c64(cutout), len(rx), grid)
        res = plan.run(d_rx, shift_start=lo, num_shifts=cnt, rows="max" if absResult else False, peak=False, cqf=not absResult)
        if absResult:
            if rel is None:  # float32 on the device -> the float64 result, widened by the download (caf_d2h_f64)
                _lib.check(_lib.load().caf_d2h_f64(out.ctypes.data, ct.c_void_p(res.row_max.ptr), ns, None), "caf_d2h_f64")
            else:
                out[:] = _host_take(res.row_max[0], rel, np.float64)
        else:
            # branch A' is sum(conj(rx) * cutout) (np.vdot order, :503): the conjugate of the engine's value
            z = res.cqf.get()[0, 0]
            np.conjugate(z if rel is None else z[rel], out=out)
        plan.close()
        return out

    d_cut = asarray(_c64(cutout).conj())
    if not outputCAF:
        out = np.zeros(ns, dtype=np.float64 if absResult else np.complex128)
        fidx = np.zeros(ns, dtype=np.uint32)
    else:
        out = np.zeros((ns, n), dtype=np.float64 if absResult else np.complex128)
    chunk = max(1, _MAX_PLANE_ELEMS // n)
    for off, start, step, count in _runs(shifts):
        for c0 in range(0, count, chunk):
            c = min(chunk, count - c0)
            sl = slice(off + c0, off + c0 + c)
            s0 = start + c0 * step
            if not outputCAF and absResult:
                q, fi, _, _ = _perdelay(d_cut, n, d_rx, len(rx), s0, step, c, False, True, True, False, False)
                out[sl], fidx[sl] = q, fi.astype(np.uint32)
            elif not outputCAF:
                _, fi, _, cc = _perdelay(d_cut, n, d_rx, len(rx), s0, step, c, False, False, True, False, True)
                fidx[sl] = fi.astype(np.uint32)
                out[sl] = cc[np.arange(c), fi]
            elif absResult:
                out[sl] = _perdelay(d_cut, n, d_rx, len(rx), s0, step, c, False, False, False, True, False)[2]
            else:
                out[sl] = _perdelay(d_cut, n, d_rx, len(rx), s0, step, c, False, False, False, False, True)[3]
    if not outputCAF:
        return out, fidx
    return out


_CZTXCORR_FORCE_ROWS = None  # tests: True / False pins cztXcorr's path


_CZT_OBJECTS = {}  # the last few CZT objects of cztXcorr's per-delay form (chirps + transform plans: milliseconds to build)


def _czt_object(n, f1, f2, step, fs):
    key = (int(n), float(f1), float(f2), float(step), float(fs))
    obj = _CZT_OBJECTS.pop(key, None)
    if obj is None:
        obj = CZTCachedGPU(n, f1, f2, step, fs)
    _CZT_OBJECTS[key] = obj  # (most recently used last)
    while len(_CZT_OBJECTS) > 4:
        _CZT_OBJECTS.pop(next(iter(_CZT_OBJECTS)))
    return obj


def _czt_rows_pay(n, k, cnt):
    """Per-delay CZT rows (3 transforms of CZTCachedGPU's nfft = next_fast_len(n + k + 1) per delay) against k hypotheses over
    overlap-save blocks of the engine's size for an n-sample template (caf_plan.hip: 16 n clipped to 2^12 .. 2^18, at least 2 n).
    (Round 4 kept rows of more than 32768 points on the engine because such calls "settled at 100 ms": that was the runtime
    pinning the 3.2 MB result array of those calls, not the rows -- csrc/caf_host.cpp; the rule is cost alone again.)"""
    nfft = next_fast_len(n + k + 1)
    rows_cost = 3.0 * cnt * nfft * np.log2(nfft)
    lb = max(min(max(int(np.ceil(np.log2(16 * n))), 12), 18), int(np.ceil(np.log2(2 * n))))
    nblk = -(-cnt // ((1 << lb) - n + 1))
    engine_cost = float(k) * nblk * (1 << lb) * lb
    return rows_cost * 2 < engine_cost


def cztXcorr(cutout, rx, f_searchMin, f_searchMax, fs, cztStep=0.1, outputCAF=False, shifts=None):
    """ref: xcorrRoutines.py:413-457.  The frequency grid is the reference's CZT grid, evaluated as explicit
    hypotheses: k = int((f2-f1)/cztStep + 1) bins, reported as f1 + i*cztStep (CZTCached.getFreq), computed
    with CZTCached's chirp rate W = (f2 - f1 + cztStep) / (k * fs) (spectralRoutines.py:239-311), i.e. at
    f1 + i*(f2 - f1 + cztStep)/k -- the same numbers whenever (f2-f1)/cztStep is an integer, and the reference's
    actual values (not its labels) when it is not."""
    cutout = np.asarray(cutout)
    rx = np.asarray(rx)
    n = cutout.size
    k = int((f_searchMax - f_searchMin) / cztStep + 1)
    f_search = np.arange(k) * cztStep + f_searchMin
    f_eval = f_searchMin + np.arange(k) * ((f_searchMax - f_searchMin + cztStep) / k)
    if shifts is None:
        shifts = np.arange(len(rx) - n + 1)
    shifts = np.asarray(shifts)
    lo, cnt, rel = _engine_range(shifts)
    d_rx = asarray(_c64(rx))
    if _czt_rows_pay(n, k, cnt) if _CZTXCORR_FORCE_ROWS is None else _CZTXCORR_FORCE_ROWS:
        # Few delays: the reference's own per-delay form (product row, CZT, norms: xcorrRoutines.py:436-452) as the kernel chain
        # of cp_fastXcorr_v2 -- sliding normalised products, ONE batched CZT, argmax per row -- instead of k hypotheses over
        # whole overlap-save blocks.  Same chirp as the engine's frequency list (CZTCached's W, see above).
        czt = _czt_object(n, f_searchMin, f_searchMax, cztStep, fs)
        assert czt.k == k
        d_pdts = multiplySlidesNormalised(asarray(_c64(cutout).conj()), d_rx, lo, cnt)
        d_spec = czt.runMany(d_pdts)  # (cnt, k) complex QF
        if outputCAF:
            out = empty((cnt, k), np.float32)
            _lib.check(_lib.load().caf_complex_magnsq(ct.c_void_p(d_spec.ptr), d_spec.size, 0, ct.c_void_p(out.ptr), 0, None))
            return out.get()[rel].astype(np.float64), f_search
        spec = d_spec.get()[rel]
        mi = np.argmax(np.abs(spec), axis=1)
        result = spec[np.arange(rel.size), mi].astype(rx.dtype if np.iscomplexobj(rx) else np.complex64)
        return result, f_search[mi].astype(np.float64)
    plan = CAFPlan(_c64(cutout), max_rx_len=len(rx), freqs_norm=f_eval / fs,
                   engine="auto" if outputCAF else "rocfft")
    if outputCAF:
        out = _host_surface(plan, d_rx, lo, cnt, rel, np.float64)
        plan.close()
        return out, f_search
    res = plan.run(d_rx, shift_start=lo, num_shifts=cnt, rows=True, peak=False, cqf=True)
    mi = res.row_arg.get()[0][rel]
    cq = res.cqf.get()[0]  # (k, cnt)
    result = cq[mi, rel].astype(rx.dtype if np.iscomplexobj(rx) else np.complex64)
    plan.close()
    return result, f_search[mi].astype(np.float64)


class _GroupEngine:
    """Shared machinery of the grouped-template classes: composite template + cached CAFPlan."""

    def _setup(self, tmpl_groups, rel_starts, lengths, autoConj, bins=None, grid=None, freqs_norm=None):
        rel_starts = np.asarray(rel_starts, dtype=np.int64)
        lengths = np.asarray(lengths, dtype=np.int64)
        self._span = int(np.max(rel_starts + lengths))
        comp = np.zeros(self._span, np.complex64)
        for g, (r, l) in enumerate(zip(rel_starts, lengths)):
            comp[r : r + l] = tmpl_groups[g][:l]
        self._comp, self._rel, self._len = comp, rel_starts, lengths
        self._autoConj, self._bins, self._grid, self._fn = autoConj, bins, grid, freqs_norm
        self._plan, self._plan_len = None, -1
        self._rows_state = None

    def _get_plan(self, rx_len):
        if self._plan is None or rx_len > self._plan_len:
            if self._plan is not None:
                self._plan.close()
            self._plan = CAFPlan(self._comp, max_rx_len=rx_len, bins=self._bins, grid=self._grid,
                                 freqs_norm=self._fn, group_starts=self._rel, group_lens=self._len,
                                 autoConj=self._autoConj)
            self._plan_len = rx_len
        return self._plan

    def _run(self, rx, shifts, **kw):
        d_rx = rx if isinstance(rx, DeviceArray) else asarray(_c64(np.asarray(rx)))  # device input stays where it is
        if kw.get("surface") and not kw.get("rows", True) and not kw.get("peak", True) and self._rows_path_pays(shifts):
            return self._run_rows(d_rx, np.asarray(shifts))
        plan = self._get_plan(d_rx.size)
        lo, cnt, rel = _engine_range(shifts)
        res = plan.run(d_rx, shift_start=lo, num_shifts=cnt, **kw)
        return res, rel

    def _surface_host(self, rx, shifts, dtype):
        """(len(shifts), F) host array of ``dtype``: the surface of ``_run(surface=True)`` brought home (see _host_surface)."""
        d_rx = rx if isinstance(rx, DeviceArray) else asarray(_c64(np.asarray(rx)))
        if self._rows_path_pays(shifts):
            res, rel = self._run_rows(d_rx, np.asarray(shifts))
            return res.surface.get()[0][rel].astype(dtype, copy=False)
        plan = self._get_plan(d_rx.size)
        lo, cnt, rel = _engine_range(shifts)
        return _host_surface(plan, d_rx, lo, cnt, rel, dtype)

    # -- the per-delay form for FEW delays over a LONG composite template ---------------------------------------
    # The hypothesis engine transforms whole overlap-save blocks of at least twice the template span per frequency:
    # the reference's benchmark_groupXcorrs.py (100 groups of 5000 samples spread over 10^6, 41 shifts, 201 CZT bins)
    # is one 2^21-point block per bin -- 8 ms.  The reference's own algorithm (xcorrRoutines.py:996-1039,
    # GroupXcorrCZT.cpp:106-329) is per delay and per group: product row, chirp-Z transform, phase of the group's
    # start, coherent sum.  For equal-length groups on a CZT grid that is ONE indexed product launch for all
    # (group, delay) rows, ONE batched CZT and one sum over the groups (caf_sum_groups_qf2).
    _czt_grid = None      # (f1, f2, binWidth, fs) when the frequency list is such a grid
    _force_rows = None    # tests: True / False pins the path

    def _rows_path_pays(self, shifts):
        if self._czt_grid is None or np.unique(self._len).size != 1:
            return False
        # (a grid whose span is not a whole number of bins: CZTCached evaluates at f1 + i (f2 - f1 + bw) / k but labels -- and
        #  phases the groups -- at f1 + i bw (spectralRoutines.py:239-311, xcorrRoutines.py:996-1039); the engine evaluates
        #  one frequency list for both.  The two forms agree only on whole-bin grids, so only those may take either.)
        f1, f2, bw, _ = self._czt_grid
        r = (f2 - f1) / bw
        if abs(r - round(r)) > 1e-9 * max(1.0, abs(r)):
            return False
        if self._force_rows is not None:
            return bool(self._force_rows)
        S, G, L = int(np.asarray(shifts).size), int(self._len.size), int(self._len[0])
        k = int(np.asarray(self._fn).size)
        nfft = next_fast_len(L + k + 1)
        rows_cost = 3.0 * S * G * nfft * np.log2(nfft)
        lo, cnt, _ = _engine_range(shifts)
        blk = 1 << int(np.ceil(np.log2(2 * self._span)))  # the engine's block for a template of this span (>= 2 N)
        nblk = -(-cnt // max(1, blk - self._span + 1))
        engine_cost = 1.0 * k * nblk * blk * np.log2(blk)
        # (templates of up to 32768 samples: the in-LDS engines, always)
        return self._span > 32768 and rows_cost * 2 < engine_cost

    def _run_rows(self, d_rx, shifts):
        f1, f2, bw, fs = self._czt_grid
        G, L, S = int(self._len.size), int(self._len[0]), int(shifts.size)
        lib = _lib.load()
        st = getattr(self, "_rows_state", None)
        if st is None:
            tm = np.stack([self._comp[r : r + L] for r in self._rel])
            if self._autoConj:
                tm = tm.conj()
            czt = CZTCachedGPU(L, f1, f2, bw, fs)
            ph = np.exp(-2j * np.pi * np.outer(self._rel.astype(np.float64), czt.getFreq()) / fs).astype(np.complex64)
            st = self._rows_state = dict(d_tm=asarray(_c64(tm)), czt=czt, d_ph=asarray(ph),
                                         ynormsq=float(np.sum(np.abs(tm.astype(np.complex128)) ** 2)))
        czt, k = st["czt"], st["czt"].k
        assert k == int(np.asarray(self._fn).size), "CZT grid and frequency list disagree"
        d_out = empty((S, k), np.float64)
        # rx energy under the groups at every delay: one moving sum of |rx|^2 (then a gather per chunk, summed over the groups on the host)
        d_msum = cupyMovingAverage(cupyComplexMagnSq(d_rx, np.float32), L, sumInstead=True)
        # the (G S, L) product matrix and the (G S, k) planes are bounded: delays in chunks of at most 2^27 matrix elements
        per = max(1, _MAX_PLANE_ELEMS // (G * max(L, k)))
        for s0 in range(0, S, per):
            sh = shifts[s0 : s0 + per]
            Sc = int(sh.size)
            starts = (sh[None, :].astype(np.int64) + self._rel[:, None]).reshape(-1)  # [g][s]
            assert starts.min() >= 0 and starts.max() + L <= d_rx.size
            d_starts = asarray(starts.astype(np.int32))
            d_len = asarray(np.full(G * Sc, L, np.int32))
            d_row = asarray(np.repeat(np.arange(G, dtype=np.int32), Sc))
            d_mul = empty((G * Sc, L), np.complex64)
            _lib.check(lib.caf_multiply_slices_indexed_rows(ct.c_void_p(d_rx.ptr), d_rx.size, ct.c_void_p(st["d_tm"].ptr), G, L,
                                                            ct.c_void_p(d_starts.ptr), ct.c_void_p(d_len.ptr), ct.c_void_p(d_row.ptr),
                                                            L, G * Sc, ct.c_void_p(d_mul.ptr), None), "caf_multiply_slices_indexed_rows")
            d_planes = czt.runMany(d_mul)  # (G Sc, k) complex64
            d_e = empty((G * Sc,), np.float32)
            d_ei = asarray((starts + L - 1).astype(np.int32))
            _lib.check(lib.caf_gather_b32(ct.c_void_p(d_msum.ptr), d_msum.size, ct.c_void_p(d_ei.ptr), G * Sc, ct.c_void_p(d_e.ptr), None),
                       "caf_gather_b32")
            d_norm = asarray(d_e.get().astype(np.float64).reshape(G, Sc).sum(axis=0))
            _lib.check(lib.caf_sum_groups_qf2(ct.c_void_p(d_planes.ptr), G, Sc, k, ct.c_void_p(st["d_ph"].ptr), ct.c_void_p(d_norm.ptr),
                                              st["ynormsq"], ct.c_void_p(d_out[s0 : s0 + Sc].ptr), None), "caf_sum_groups_qf2")
        return _RowsResult(d_out), np.arange(S)


class _RowsResult:
    """What _GroupEngine._run hands back from the per-delay form: ``surface.get()`` -> float32 (1, S, k) like a CAFResult's."""

    class _Surface:
        def __init__(self, d):
            self._d = d

        def get(self):
            return self._d.get().astype(np.float32)[None]

    def __init__(self, d_out):
        self.surface = self._Surface(d_out)


class GroupXcorr(_GroupEngine):
    """ref: xcorrRoutines.py:852-954.  Composite template (groups), explicit frequency list.
    ``xcorr`` returns (QF^2 float64[S], peak frequency in Hz float64[S])."""

    def __init__(self, y, starts, lengths, freqs, fs, autoConj=True, autoZeroStarts=True):
        y = np.asarray(y)
        starts = np.asarray(starts)
        lengths = np.asarray(lengths)
        assert starts.size == lengths.size
        self.starts = starts - starts[0] if autoZeroStarts else starts
        self.lengths = lengths
        self.numGroups = starts.size
        self.freqs = np.asarray(freqs, dtype=np.float64)
        self.fs = fs
        groups = [_c64(y[s : s + l]) for s, l in zip(starts, lengths)]
        self.yconcat = np.hstack([g.conj() if autoConj else g for g in groups])
        self.yconcatNormSq = float(np.sum(np.abs(self.yconcat.astype(np.complex128)) ** 2))
        self._setup(groups, self.starts - self.starts[0], lengths, autoConj, freqs_norm=self.freqs / fs)
        self._first = int(self.starts[0])
        # a uniformly spaced frequency list is a CZT grid: few shifts over a long composite template can then take the
        # per-delay form of _GroupEngine (the maximum over the frequencies is taken from its (S, k) plane)
        if self.freqs.size >= 2:
            bw = float(self.freqs[1] - self.freqs[0])
            if bw > 0 and np.allclose(np.diff(self.freqs), bw, rtol=0, atol=1e-9 * max(1.0, abs(bw))):
                f1, f2 = float(self.freqs[0]), float(self.freqs[-1])
                if int((f2 - f1) / bw + 1) == self.freqs.size:
                    self._czt_grid = (f1, f2, bw, fs)

    def xcorr(self, rx, shifts=None):
        rx = np.asarray(rx)
        if shifts is None:
            shifts = np.arange(len(rx) - (self.starts[-1] + self.lengths[-1]) + 1)
        else:
            shifts = np.asarray(shifts)
            assert shifts[-1] + self.starts[-1] + self.lengths[-1] < rx.size
        if self._rows_path_pays(shifts + self._first):
            d_rx = asarray(_c64(rx))
            res, _ = self._run_rows(d_rx, shifts + self._first)
            plane = res.surface.get()[0]
            mi = np.argmax(plane, axis=1)
            return plane[np.arange(mi.size), mi].astype(np.float64), self.freqs[mi]
        res, rel = self._run(rx, shifts + self._first, rows=True, peak=False)
        xc = _host_take(res.row_max[0], rel, np.float64)
        return xc, self.freqs[_host_take(res.row_arg[0], rel, np.int32)]


class GroupXcorrCZT(_GroupEngine):
    """ref: xcorrRoutines.py:957-1039.  ``xcorr`` returns (QF^2 float64[S, k], cztFreq)."""

    def __init__(self, y, starts, lengths, f1, f2, binWidth, fs, autoConj=True, autoZeroStarts=True):
        y = np.asarray(y)
        starts = np.asarray(starts)
        lengths = np.asarray(lengths)
        assert starts.size == lengths.size
        self.starts = starts - starts[0] if autoZeroStarts else starts
        self.lengths = lengths
        self.numGroups = starts.size
        self.fs = fs
        self.f1, self.f2, self.binWidth = f1, f2, binWidth
        self.maxLength = int(np.max(lengths))
        groups = [_c64(y[s : s + l]) for s, l in zip(starts, lengths)]
        self.ystackNormSq = float(sum(np.sum(np.abs(g.astype(np.complex128)) ** 2) for g in groups))
        self._k = int((f2 - f1) / binWidth + 1)
        self._freq = np.arange(f1, f2 + binWidth / 2, binWidth)
        # (evaluated where CZTCached evaluates -- f1 + i (f2 - f1 + binWidth) / k, the labels whenever the span is a whole
        #  number of bins --, like cztXcorr above; reported at the labels)
        f_eval = f1 + np.arange(self._k) * ((f2 - f1 + binWidth) / self._k)
        self._setup(groups, self.starts - self.starts[0], lengths, autoConj, freqs_norm=f_eval / fs)
        self._czt_grid = (f1, f2, binWidth, fs)
        self._first = int(self.starts[0])

    def xcorr(self, rx, shifts=None):
        rx = np.asarray(rx)
        if shifts is None:
            shifts = np.arange(len(rx) - (self.starts[-1] + self.lengths[-1]) + 1)
        else:
            shifts = np.asarray(shifts)
            assert shifts[-1] + self.starts[-1] + self.lengths[-1] < rx.size
        return self._surface_host(rx, shifts + self._first, np.float64), self._freq


class GroupXcorrFFT(_GroupEngine):
    """ref: xcorrRoutines.py:1047-1262.  Equal-length groups on the makeFreq(fftlen, fs) grid.
    ``xcorr`` -> (float64[S], uint32[S]) or float64[S, fftlen]; ``xcorrGPU`` takes a DeviceArray."""

    def __init__(self, ygroups, starts, fs, autoConj=True, fftlen=None, autoZeroStarts=True):
        ygroups = np.asarray(ygroups)
        starts = np.asarray(starts)
        assert starts.size == ygroups.shape[0]
        self.starts = starts - starts[0] if autoZeroStarts else starts
        self.numGroups = starts.size
        self.fs = fs
        self.ygroupLen = ygroups.shape[1]
        self.fftlen = self.ygroupLen if fftlen is None else int(fftlen)
        self.ygroupNormSq = float(np.sum(np.abs(ygroups.astype(np.complex128)) ** 2))
        self.ygroups = ygroups.conj() if autoConj else ygroups
        self.fftfreq = makeFreq(self.fftlen, self.fs)
        groups = [_c64(g) for g in ygroups]
        lens = np.full(self.numGroups, self.ygroupLen)
        pow2 = (self.fftlen & (self.fftlen - 1)) == 0
        if pow2:
            self._setup(groups, self.starts - self.starts[0], lens, autoConj, bins=np.arange(self.fftlen),
                        grid=self.fftlen)
        else:
            self._setup(groups, self.starts - self.starts[0], lens, autoConj,
                        freqs_norm=np.arange(self.fftlen) / self.fftlen)
        self._first = int(self.starts[0])

    def _default_shifts(self, n):
        return np.arange(n - (self.starts[-1] + self.fftlen) + 1)

    def _shifts_for(self, rx, shifts):
        if shifts is None:
            return self._default_shifts(rx.size)
        shifts = np.asarray(shifts)
        assert shifts[-1] + self.starts[-1] + self.fftlen < rx.size
        return shifts

    def xcorr(self, rx, shifts=None, flattenToTime=True):
        rx = rx if isinstance(rx, DeviceArray) else np.asarray(rx)
        shifts = self._shifts_for(rx, shifts)
        if flattenToTime:
            res, rel = self._run(rx, shifts + self._first, rows=True, peak=False)
            return _host_take(res.row_max[0], rel, np.float64), _host_take(res.row_arg[0], rel, np.uint32)
        return self._surface_host(rx, shifts + self._first, np.float64)

    def xcorrThreads(self, rx, shifts=None, NUM_THREADS=4):
        return self.xcorr(rx, shifts, flattenToTime=False)

    def xcorrGPU(self, rx, shifts=None, flattenToTime=True):
        """ref: xcorrRoutines.py:1191-1262: device rx in, device results out -- (float64[S], uint32[S]) or
        float64[S, fftlen] -- and nothing crosses PCIe but the shift list."""
        requireDeviceArray(rx)
        shifts = self._shifts_for(rx, shifts)
        if flattenToTime:
            res, rel = self._run(rx, shifts + self._first, rows=True, peak=False)
            return _take_f64(res.row_max[0], rel), _take_u32(res.row_arg[0], rel)
        res, rel = self._run(rx, shifts + self._first, surface=True, rows=False, peak=False)
        S, F = res.surface.shape[1], res.surface.shape[2]
        # rows rel of the (S, F) float32 surface as float64: flat element indices (identity when rel is 0..S-1)
        rel = np.asarray(rel)
        contiguous = rel.size == S and _dev_index(rel, S) is None
        flat = np.arange(S * F) if contiguous else (rel[:, None] * F + np.arange(F)[None, :]).reshape(-1)
        return _take_f64(res.surface.reshape(S * F), flat).reshape(rel.size, F)


# ------------------------------------------------------------------------------------------
# device-array (cupy-signature) entry points
# ------------------------------------------------------------------------------------------
class TemplateCrossCorrelator:
    """ref: xcorrRoutines.py:277-371.  T templates, no frequency scan; returns QF (not QF^2):
    complex64 (T, M-L+1), or with returnMax (float32 QF[M-L+1], int64 templateIdx[M-L+1]).

    The complex plane comes from the one-launch in-LDS engine for templates of up to 8192 samples (its FFT work items
    write the normalised complex rows themselves), from the rocFFT engine beyond.  ``returnMax=True`` is bit-for-bit
    the column max / argmax of that complex output, the property the reference's unit test asserts
    (xcorrRoutines.py:2229-2233).

    ``fastMax`` (not upstream, default off): ``correlate(returnMax=True)`` then never forms the complex plane at all
    (per-template QF^2 traces from the same engine, then the maximum of their square roots): it saves the 8.6 GB
    round trip of the plane at config C3 and agrees with the default to float32 rounding."""

    def __init__(self, templates, inputSize, fastMax=False):
        self._fastMax = bool(fastMax)
        self._inputSize = int(inputSize)
        requireDeviceArray(templates)
        if not templates.ndim == 2:
            raise ValueError("Templates must be a 2D array; 1 row for 1 template.")
        self._templateOrigLength = templates.shape[1]
        tm = templates.get()
        self._templateNorms = asarray(np.sqrt(np.sum(np.abs(tm.astype(np.complex128)) ** 2, axis=1)).astype(np.float32))
        self._tm = _c64(tm)
        self._grid = 1 << int(np.ceil(np.log2(max(self._templateOrigLength, 2))))
        self._plan = None      # complex QF (rocfft engine), built on first use
        self._plan_max = None  # per-template QF^2 traces (one-launch engine) for returnMax, built on first use

    def correlate(self, x, returnMax=False):
        requireDeviceArray(x)
        if x.ndim != 1 or x.size != self._inputSize:
            raise ValueError("x must be 1D of length %d" % self._inputSize)
        requireDtype(np.complex64, x)
        T = self._tm.shape[0]
        S = self._inputSize - self._templateOrigLength + 1
        if returnMax and self._fastMax and self._templateOrigLength <= 8192:
            # only |QF| and the best template per delay are wanted: per-template QF^2 traces from the one-launch
            # engine (no complex plane), then the maximum of their square roots over the templates
            if self._plan_max is None:
                self._plan_max = CAFPlan(self._tm, max_rx_len=self._inputSize, bins=[0], grid=self._grid)
            res = self._plan_max.run(x, surface=False, rows="max", peak=False)
            qf = empty(S, np.float32)
            ti = empty(S, np.int64)
            _lib.check(_lib.load().caf_colmax_sqrt(ct.c_void_p(res.row_max.ptr), T, S, ct.c_void_p(qf.ptr),
                                                   ct.c_void_p(ti.ptr), None))
            return qf, ti
        if self._plan is None:
            # templates of up to 8192 samples: the one-launch in-LDS engine writes the complex rows itself (one transform
            # = one row segment; config C3: 3.x ms for 8.6 GB of output); longer ones: multiply -> rocFFT -> normalise
            self._plan = _complex_qf_plan(self._tm, self._inputSize, self._grid)
        res = self._plan.run(x, rows=False, peak=False, cqf=True)
        nout = res.cqf.reshape(T, S)
        if not returnMax:
            return nout
        qf = empty(S, np.float32)
        ti = empty(S, np.int64)  # cp.argmax's dtype, written by the kernel
        _lib.check(_lib.load().caf_colmax_abs(ct.c_void_p(nout.ptr), T, S, ct.c_void_p(qf.ptr), ct.c_void_p(ti.ptr), 1,
                                              None))
        return qf, ti


def cp_fastXcorr(cutout, rx, freqsearch=True, outputCAF=False, shifts=None, absResult=True, BATCH=1024, copyToCpu=True):
    """ref: xcorrRoutines.py:29-167.  Only the frequency-scanning / flattened / abs branch exists
    upstream (the others print "Not implemented.").  Returns (float64 QF^2[S], uint32 bin[S])."""
    if isinstance(cutout, DeviceArray):
        cutout = cutout.get()
    d_rx = rx if isinstance(rx, DeviceArray) else asarray(_c64(rx))
    if np.dtype(cutout.dtype) != d_rx.dtype:
        raise Exception("Cutout and Rx must be same type, please cast one of them manually.")
    n = len(cutout)
    if shifts is None:
        shifts = np.arange(d_rx.size - n + 1)
    if not freqsearch or outputCAF or not absResult:
        print("Not implemented.")
        return None
    shifts = np.asarray(shifts)
    d_cut = asarray(_c64(cutout).conj())
    if not copyToCpu:
        # results stay on the device (float64 / uint32 like the reference's d_result / d_freqlist): every run of
        # shifts writes its slice of the two output arrays
        d_out, d_fidx = empty(len(shifts), np.float64), empty(len(shifts), np.uint32)
        for off, start, step, count in _runs(shifts):
            q, fi, _, _ = _perdelay(d_cut, n, d_rx, d_rx.size, start, step, count, False, True, True, False, False,
                                    device_out=True)
            ident = np.arange(count)
            _take_f64(q, ident, out=d_out[off : off + count])
            _take_u32(fi, ident, out=d_fidx[off : off + count])
        return d_out, d_fidx
    out = np.zeros(len(shifts), np.float64)
    fidx = np.zeros(len(shifts), np.uint32)
    for off, start, step, count in _runs(shifts):
        q, fi, _, _ = _perdelay(d_cut, n, d_rx, d_rx.size, start, step, count, False, True, True, False, False)
        out[off : off + count], fidx[off : off + count] = q, fi.astype(np.uint32)
    return out, fidx


def _one_kernel_cutout(n):
    """Cutout lengths that caf_xcorr_perdelay serves with one fused kernel (the library's own rule: powers of two, powers
    of ten, 2^a 3^b 5^c 7^d lengths with a mixed-radix plan)."""
    return bool(_lib.load().caf_xcorr_perdelay_one_kernel(int(n)))


def cp_fastXcorr_v2(cutout, rx, startIdx=0, idxlen=None, THREADS_PER_BLOCK=32, numSlidesPerBlk=None, cztObj=None,
                    flattenCAF=False, BATCH=None):
    """ref: xcorrRoutines.py:169-274.  Kernel chain on device arrays: sliding normalised product
    (template NOT conjugated here, as upstream) -> row FFT or CZT -> argmax | |.|^2.
    Returns (uint32 freqIdx, float32 qf2) -- order swapped vs v1, as upstream -- or the float32
    (idxlen, N | k) plane.  Every batch writes its own rows (the upstream multi-batch indexing slip
    at :263-265 is not reproduced)."""
    requireDeviceArray(cutout)
    requireDeviceArray(rx)
    if idxlen is None:
        idxlen = rx.size - cutout.size - startIdx + 1
    if cztObj is not None and cztObj.m != cutout.size:
        raise ValueError("CZT object input length doesn't match the cutout array size")
    if BATCH is None:
        BATCH = idxlen
    ncols = cutout.size if cztObj is None else cztObj.k
    BATCH = max(1, min(int(BATCH), max(1, _MAX_PLANE_ELEMS // max(cutout.size, ncols))))
    if flattenCAF:
        d_freqIdx = empty(idxlen, np.uint32)
        d_qf2 = empty(idxlen, np.float32)
    else:
        d_out = empty((idxlen, ncols), np.float32)
    lib = _lib.load()
    n = int(cutout.size)
    if (flattenCAF and cztObj is None and _one_kernel_cutout(n) and startIdx >= 0
            and startIdx + idxlen - 1 + n <= rx.size):
        # power-of-two cutout, 100 / 1000 / 10000 samples (radix-10 passes) or any other 2^a 3^b 5^c 7^d length up to 16200
        # (mixed-radix passes), every window inside rx: the whole chain (product, row transform, |.|^2, argmax, both norms)
        # is ONE kernel (caf_perdelay.hip, caf_perdelay_mr.hip) -- no (idxlen, N) matrix, no batches
        _lib.check(lib.caf_xcorr_perdelay(ct.c_void_p(cutout.ptr), n, ct.c_void_p(rx.ptr), rx.size, int(startIdx), 1,
                                          int(idxlen), 0, ct.c_void_p(d_qf2.ptr), ct.c_void_p(d_freqIdx.ptr), None, None,
                                          0, None), "caf_xcorr_perdelay")
        return d_freqIdx, d_qf2
    fi = 0
    while fi < idxlen:
        nb = min(BATCH, idxlen - fi)
        d_pdts = multiplySlidesNormalised(cutout, rx, startIdx + fi, nb)
        d_spec = fftRows(d_pdts, out=d_pdts) if cztObj is None else cztObj.runMany(d_pdts)
        if flattenCAF:
            cupyArgmaxAbsRows_complex64(d_spec, d_argmax=d_freqIdx[fi : fi + nb], d_max=d_qf2[fi : fi + nb],
                                        returnMaxValues=True, useNormSqInstead=True)
        else:
            _lib.check(lib.caf_complex_magnsq(ct.c_void_p(d_spec.ptr), d_spec.size, 0, ct.c_void_p(d_out[fi : fi + nb].ptr),
                                              0, None))
        fi += nb  # (no synchronisation per batch: the scratch of a batch is recycled in stream order)
    if flattenCAF:
        return d_freqIdx, d_qf2
    return d_out


# ------------------------------------------------------------------------------------------
# native-class signatures (Cython / pybind twins upstream)
# ------------------------------------------------------------------------------------------
class CyIppXcorrFFT:
    """ref: cython_ext/CyIppXcorrFFT/CyIppXcorrFFT.pyx:6-83, IppXcorrFFT.cpp:94-194.
    ``xcorr(rx, startIdx, endIdx, step)`` -> (float32 QF^2, int32 bin); delays whose window leaves
    rx give (0.0, 0) instead of an error.  ``num_threads`` is accepted and ignored (one GPU)."""

    def __init__(self, cutout, num_threads=1, autoConj=True):
        cutout = np.asarray(cutout)
        if cutout.dtype != np.complex64 or cutout.ndim != 1:
            raise ValueError("Buffer dtype mismatch, expected 1-D complex64 cutout")
        self._n = cutout.size
        self._d_cut = asarray(cutout.conj() if autoConj else cutout)
        self._last = None

    def xcorr(self, rx, startIdx, endIdx, step):
        rx = np.asarray(rx)
        if rx.dtype != np.complex64 or rx.ndim != 1:
            raise ValueError("Buffer dtype mismatch, expected 1-D complex64 rx")
        length = len(np.arange(startIdx, endIdx, step))
        if length == 0:
            return np.zeros(0, np.float32), np.zeros(0, np.int32)
        q, fi, _, _ = _perdelay(self._d_cut, self._n, asarray(rx), rx.size, startIdx, step, length, True, True, True,
                                False, False)
        self._last = (q, fi)
        return q, fi

    def results(self):
        return self._last


class CyGroupXcorrFFT(_GroupEngine):
    """ref: cython_ext/CyGroupXcorrFFT/CyGroupXcorrFFT.pyx:6-65, GroupXcorrFFT.cpp:3-203.
    ``xcorr(rx, shifts, NUM_THREADS)`` -> float32 (S, fftlen) (always the full plane)."""

    def __init__(self, ygroups, offsets, fs, fftlen=-1, autoConj=True):
        ygroups = np.asarray(ygroups)
        offsets = np.asarray(offsets)
        if ygroups.dtype != np.complex64 or ygroups.ndim != 2:
            raise ValueError("ygroups must be 2-D complex64")
        if offsets.dtype != np.int32:
            raise ValueError("offsets must be int32")
        L = ygroups.shape[1]
        self.fftlen = L if fftlen == -1 else int(fftlen)
        if self.fftlen < L:
            raise ValueError("INVALID_FFTLEN: fftlen must be >= group length")
        rel = offsets.astype(np.int64) - int(offsets[0])
        lens = np.full(ygroups.shape[0], L)
        if (self.fftlen & (self.fftlen - 1)) == 0:
            self._setup([g for g in ygroups], rel, lens, autoConj, bins=np.arange(self.fftlen), grid=self.fftlen)
        else:
            self._setup([g for g in ygroups], rel, lens, autoConj, freqs_norm=np.arange(self.fftlen) / self.fftlen)

    def xcorr(self, rx, shifts, NUM_THREADS=1):
        rx = np.asarray(rx)
        shifts = np.asarray(shifts)
        if rx.dtype != np.complex64 or shifts.dtype != np.int32:
            raise ValueError("rx must be complex64 and shifts int32")
        return self._surface_host(rx, shifts, np.float32)


class pbIppGroupXcorrCZT(_GroupEngine):
    """ref: pybinds/ippGroupXcorrCZT (pbGroupXcorrCZT.cpp:7-40, GroupXcorrCZT.cpp:4-375).
    addGroup / addGroupsFromArray / resetGroups / xcorr(x, shiftStart, shiftStep, numShifts)
    -> float32 (numShifts, k); the reference's range_error / invalid_argument become
    IndexError / ValueError."""

    def __init__(self, maxlen, f1, f2, fstep, fs, NUM_THREADS=1):
        if NUM_THREADS < 1:
            raise ValueError("Number of threads must be greater than 0")
        self._N, self._f1, self._f2, self._fstep, self._fs = int(maxlen), f1, f2, fstep, fs
        self._k = int((f2 - f1) / fstep + 1)
        self._threads = NUM_THREADS
        self.resetGroups()

    def getNumThreads(self):
        return self._threads

    def resetGroups(self):
        self._gstarts, self._groups = [], []
        self._plan, self._plan_len = None, -1
        self._ready = False  # the composite template (and what either engine keeps for it) is built at the next xcorr

    def addGroup(self, start, group, autoConj=True):
        group = np.asarray(group)
        if group.dtype != np.complex64:
            raise TypeError("group must be complex64")
        length = group.size
        for gs, g in zip(self._gstarts, self._groups):
            ge = gs + g.size
            if gs <= start < ge:
                raise IndexError("Group start overlaps with existing group! [%d,%d)" % (gs, ge))
            if gs <= start + length < ge:
                raise IndexError("Group end overlaps with existing group! [%d,%d)" % (gs, ge))
        if length > self._N:
            raise IndexError("Length of group exceeds maximum length")
        self._gstarts.append(int(start))
        self._groups.append(group.conj() if autoConj else group.copy())
        self._plan = None
        self._ready = False

    def addGroupsFromArray(self, starts, lengths, arr, autoConj=True):
        starts = np.asarray(starts)
        lengths = np.asarray(lengths)
        m = int(starts.min())
        for s, l in zip(starts, lengths):
            self.addGroup(int(s) - m, np.asarray(arr)[s : s + l], autoConj)

    def xcorr(self, x, shiftStart, shiftStep, numShifts):
        x = np.asarray(x)
        if x.dtype != np.complex64:
            raise TypeError("x must be complex64")
        if shiftStep < 0:
            raise ValueError("shiftStep cannot be negative")
        if not self._groups:
            raise IndexError("No groups have been defined!")
        for gs, g in zip(self._gstarts, self._groups):
            xi = shiftStart + gs
            if xi < 0:
                raise IndexError("Shifts accesses negative indices!")
            if xi + numShifts * shiftStep + g.size >= x.size:
                raise IndexError("Input length is insufficient for search range!")
        if not self._ready:
            order = np.argsort(self._gstarts)
            gst = np.asarray(self._gstarts)[order]
            lens = np.asarray([self._groups[i].size for i in order])
            freqs = (self._f1 + np.arange(self._k) * self._fstep) / self._fs
            # stored groups are already conjugated when autoConj was requested
            self._base = int(gst[0])
            self._setup([self._groups[i] for i in order], gst - gst[0], lens, False, freqs_norm=freqs)
            self._czt_grid = (self._f1, self._f2, self._fstep, self._fs)
            self._ready = True
        shifts = shiftStart + self._base + shiftStep * np.arange(numShifts)
        return self._surface_host(x, shifts, np.float32)


# ------------------------------------------------------------------------------------------
# small helpers of the reference API
# ------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------
# sub-sample refinement after the peak
# ------------------------------------------------------------------------------------------
def makeTimeScanSteervec(td_scan_range, fs, siglen):
    """ref: xcorrRoutines.py:670-676 (complex128 (len(td_scan_range), siglen) host matrix)."""
    sigFreq = makeFreq(siglen, fs)
    return np.exp(1j * 2 * np.pi * sigFreq * np.asarray(td_scan_range).reshape((-1, 1)))


def _p(a):
    return ct.c_void_p(a.ptr)


def _mul_conj(d_a, d_b):
    out = empty(d_a.shape, np.complex64)
    _lib.check(_lib.load().caf_mul_conj(_p(d_a), _p(d_b), d_a.size, _p(out), None), "caf_mul_conj")
    return out


def _steer_dot(d_vec, d_steer, scale=1.0):
    """scale * sum_k vec[k] * conj(steer[r, k]) for every row r, float64 accumulation; complex128 host result."""
    rows, n = d_steer.shape
    out = empty((rows,), np.complex128)
    _lib.check(_lib.load().caf_steer_dot(_p(d_vec), _p(d_steer), rows, n, float(scale), _p(out), None), "caf_steer_dot")
    return out.get()


def _dev_norm(d_x):
    """||x|| of a complex64 device vector (|x|^2 -> moving sum over the whole length, float64 accumulation)."""
    s = cupyMovingAverage(cupyComplexMagnSq(d_x, np.float32), d_x.size, sumInstead=True)
    return float(np.sqrt(np.float64(s[d_x.size - 1 : d_x.size].get()[0])))


def _time_scan(d_x, d_y, d_steer, freq_mask):
    x_fft, y_fft = fftRows(d_x), fftRows(d_y)
    rx_vec = _mul_conj(x_fft, y_fft)
    if freq_mask is not None:
        rx_vec = _mul_conj(rx_vec, asarray(freq_mask.astype(np.complex64)))
    return _steer_dot(rx_vec, d_steer, 1.0 / (_dev_norm(x_fft) * _dev_norm(y_fft)))


def fineFreqTimeSearch(x_aligned, y_aligned, fineRes, freqfound, freqRes, fs, td_scan_range, steeringvec=None,
                       td_scan_freqBounds=None):
    """ref: xcorrRoutines.py:583-667.  Fine frequency (successive grids ``fineRes`` around ``freqfound``) and
    then sub-sample time alignment by a steering-vector scan of the cross spectrum.  Returns
    (finefreqfound or None, timediff, cost_vec complex128).  Device flow: y * conj(x) -> steering dot products
    (``caf_steer_dot``) for the frequency grids; FFT rows -> x_fft * conj(y_fft) (``caf_mul_conj``) -> steering
    dot products for the time scan."""
    x = _c64(np.asarray(x_aligned))
    y = _c64(np.asarray(y_aligned))
    n = x.size
    d_x, d_y = asarray(x), asarray(y)
    if len(fineRes) > 0:
        d_yx = _mul_conj(d_y, d_x)  # conj(precomputed), precomputed = y.conj() * x (:622)
        for i in range(len(fineRes)):
            fineFreq = np.arange(freqfound - freqRes, freqfound + freqRes, fineRes[i])
            fineshifts = np.exp(1j * 2 * np.pi * -fineFreq.reshape((-1, 1)) * np.arange(n) / fs)
            pp = _steer_dot(d_yx, asarray(np.conj(fineshifts)))  # = np.vdot(precomputed, fineshifts[j])
            fineFreq_ind = int(np.argmax(np.abs(pp)))
            freqfound = fineFreq[fineFreq_ind]
        finefreqfound = freqfound
        d_x = _mul_conj(d_x, asarray(_c64(np.conj(fineshifts[fineFreq_ind]))))  # x * fineshifts[ind]
    else:
        finefreqfound = None
    td_scan_range = np.asarray(td_scan_range)
    if steeringvec is None:
        steeringvec = makeTimeScanSteervec(td_scan_range, fs, n)
    mask = None
    if td_scan_freqBounds is not None:
        freqvec = makeFreq(n, fs)
        mask = ~np.logical_or(freqvec < td_scan_freqBounds[0], freqvec >= td_scan_freqBounds[1])
    cost_vec = _time_scan(d_x, d_y, asarray(np.ascontiguousarray(steeringvec, dtype=np.complex128)), mask)
    idx_td = int(np.argmax(np.abs(cost_vec)))
    return finefreqfound, td_scan_range[idx_td], cost_vec


class GenXcorr:
    """ref: xcorrRoutines.py:679-719.  The time-scan half of ``fineFreqTimeSearch`` with the steering matrix
    built (and here: uploaded) once."""

    def __init__(self, td_scan_range, fs, siglen):
        self.td_scan_range = np.asarray(td_scan_range)
        self.fs = fs
        self.sigFreq = makeFreq(siglen, fs)
        self.steeringvec = self._makeTimeScanSteervec()
        self._d_steer = asarray(np.ascontiguousarray(self.steeringvec, dtype=np.complex128))
        self.td_scan_freqBounds = None

    def _makeTimeScanSteervec(self):
        return np.exp(1j * 2 * np.pi * self.sigFreq * self.td_scan_range.reshape((-1, 1)))

    def setTDscan_freqBounds(self, td_scan_freqBounds):
        self.td_scan_freqBounds = td_scan_freqBounds

    def xcorr(self, x, y):
        mask = None
        if self.td_scan_freqBounds is not None:
            mask = ~np.logical_or(self.sigFreq < self.td_scan_freqBounds[0], self.sigFreq >= self.td_scan_freqBounds[1])
        cost_vec = _time_scan(asarray(_c64(np.asarray(x))), asarray(_c64(np.asarray(y))), self._d_steer, mask)
        return self.td_scan_range[int(np.argmax(np.abs(cost_vec)))], cost_vec


class GroupXcorrGPU(GroupXcorr):
    """ref: xcorrRoutines.py:1897-2058.  ``GroupXcorr`` without the autoConj / autoZeroStarts options;
    ``xcorr`` as the parent, ``xcorrKernel`` returns (QF^2 float32[S], frequency INDEX int32[S])."""

    def __init__(self, y, starts, lengths, freqs, fs):
        super().__init__(y, starts, lengths, freqs, fs)

    def xcorrKernel(self, rx, shifts, numShiftsPerBlk=2, verbTiming=False):
        rx = np.asarray(rx)
        shifts = np.asarray(shifts)
        assert shifts.size % numShiftsPerBlk == 0
        res, rel = self._run(rx, shifts + self._first, rows=True, peak=False)
        return _host_take(res.row_max[0], rel, np.float32), _host_take(res.row_arg[0], rel, np.int32)


class GroupXcorrCZT_Permutations:
    """ref: xcorrRoutines.py:1264-1690.  Every template (one of several candidates per group, all of one
    length) is correlated ONCE into a complex (shifts, k) CZT plane; ``getCAF`` then combines one plane per
    group into the CAF of that permutation, so P1 x P2 x ... permutations cost P1 + P2 + ... correlations.

    Device flow per template: product rows x[shift + start : +L] * template (``caf_multiply_slices_indexed_rows``)
    -> Bluestein CZT rows (``caf_czt_run_many``, the group's start phase folded into the output chirp);
    rx group energies from one moving sum of |rx|^2; ``getCAF`` = ``caf_sum_planes_qf2``.
    The planes stay on the device (``d_xcTemplates`` complex64 (T, S, k)); ``xcTemplates`` / ``rxgroupNormSq``
    are host copies for callers that read the reference's attributes."""

    def __init__(self, ygroups, ygroupIdxs, groupStarts, f1, f2, binWidth, fs, autoConj=True):
        ygroups = np.asarray(ygroups)
        ygroupIdxs = np.asarray(ygroupIdxs)
        groupStarts = np.asarray(groupStarts)
        assert ygroups.shape[0] == ygroupIdxs.size
        assert np.unique(ygroupIdxs).size == groupStarts.size
        self.numTemplates = ygroupIdxs.size
        self.numGroups = groupStarts.size
        assert np.all(np.sort(np.unique(ygroupIdxs)) == np.arange(self.numGroups))
        self.groupStarts = groupStarts
        self.ygroupIdxs = ygroupIdxs
        self.fs = fs
        self.length = ygroups.shape[1]
        self.f1, self.f2, self.binWidth = f1, f2, binWidth
        self.ygroups = ygroups.conj() if autoConj else ygroups
        self.ygroupsEnergy = np.linalg.norm(self.ygroups, axis=1) ** 2
        self._d_ygroups = asarray(_c64(self.ygroups))
        self.d_xcTemplates = None
        self.d_rxgroupNormSq = None

    # -- shared device path ------------------------------------------------------------------
    def _correlate(self, d_rx, shifts):
        shifts = np.asarray(shifts)
        L, T, G = self.length, self.numTemplates, self.numGroups
        S = shifts.size
        cztFreq = np.arange(self.f1, self.f2 + self.binWidth / 2, self.binWidth)
        czts = []
        for g in range(G):  # one CZT object per group: output chirp * exp(-j 2 pi f start_g / fs)
            c = CZTCachedGPU(L, self.f1, self.f2, self.binWidth, self.fs)
            ph = np.exp(-2j * np.pi * c.getFreq() * float(self.groupStarts[g]) / self.fs)
            c._d_wws = asarray((c._ww64[c.m - 1 : c.m + c.k - 1] * ph).astype(np.complex64))
            czts.append(c)
        k = czts[0].k
        # sliding energy of rx once, on the device: E[d] = sum |rx[d : d+L]|^2 (causal moving sum, so
        # E[d] = msum[d + L - 1]); the (G, S) energies of the wanted shifts are gathered there too -- only the
        # per-group start indices (S int32 each) are uploaded, nothing of rx's length comes back
        d_msum = cupyMovingAverage(cupyComplexMagnSq(d_rx, np.float32), L, sumInstead=True)
        lib = _lib.load()
        self.d_rxgroupNormSq = empty((G, S), np.float32)
        d_starts = []
        for g in range(G):
            d_starts.append(asarray((shifts + int(self.groupStarts[g])).astype(np.int32)))
            d_e = asarray((shifts + int(self.groupStarts[g]) + L - 1).astype(np.int32))
            _lib.check(lib.caf_gather_b32(ct.c_void_p(d_msum.ptr), d_msum.size, ct.c_void_p(d_e.ptr), S,
                                          ct.c_void_p(self.d_rxgroupNormSq[g].ptr), None), "caf_gather_b32")
        self._rxgroupNormSq_host = None  # (G, S) float64 host copy, read back on first use
        self.d_xcTemplates = empty((T, S, k), np.complex64)
        d_len = asarray(np.full(S, L, np.int32))
        d_row0 = zeros(S, np.int32)  # every slice multiplies row 0 of the one-template view passed below
        d_mul = empty((S, L), np.complex64)
        for t in range(T):
            g = int(self.ygroupIdxs[t])
            # multiplySlicesOptimistically (cupyExtensions.py:405-488) without its per-call host check of the slice
            # lengths: they are all L by construction
            _lib.check(lib.caf_multiply_slices_indexed_rows(ct.c_void_p(d_rx.ptr), d_rx.size,
                                                            ct.c_void_p(self._d_ygroups[t].ptr), 1, L,
                                                            ct.c_void_p(d_starts[g].ptr), ct.c_void_p(d_len.ptr),
                                                            ct.c_void_p(d_row0.ptr), L, S, ct.c_void_p(d_mul.ptr), None),
                       "caf_multiply_slices_indexed_rows")
            czts[g].runMany(d_mul, out=self.d_xcTemplates[t])
        self._shape = (S, k)
        return cztFreq

    @property
    def _rxgroupNormSq(self):
        if self._rxgroupNormSq_host is None:
            self._rxgroupNormSq_host = self.d_rxgroupNormSq.get().astype(np.float64)
        return self._rxgroupNormSq_host

    def _check_shifts(self, n, shifts):
        if shifts is None:
            return np.arange(n - (self.groupStarts[-1] + self.length) + 1)
        shifts = np.asarray(shifts)
        assert shifts[-1] + self.groupStarts[-1] + self.length < n
        return shifts

    # -- reference entry points --------------------------------------------------------------
    def xcorrGPU(self, rx, shifts, batchSz=32):
        requireDeviceArray(rx)
        requireDtype(np.complex64, rx)
        return self._correlate(rx, self._check_shifts(rx.size, shifts))

    def xcorr(self, rx, shifts=None, numThreads=1):
        rx = np.asarray(rx)
        return self._correlate(asarray(_c64(rx)), self._check_shifts(rx.size, shifts))

    @property
    def xcTemplates(self):
        return self.d_xcTemplates.get().astype(np.complex128)

    @property
    def rxgroupNormSq(self):
        return self._rxgroupNormSq

    def _select(self, templateIdx):
        templateIdx = np.asarray(templateIdx)
        assert templateIdx.size == self.numGroups
        sel = np.array([np.argwhere(self.ygroupIdxs == g)[templateIdx[g]][0] for g in range(self.numGroups)], np.int32)
        return sel, self._rxgroupNormSq.sum(axis=0), float(np.sum(self.ygroupsEnergy[sel]))

    def getCAF_GPU(self, templateIdx):
        sel, rxnormsq, ynormsq = self._select(templateIdx)
        S, k = self._shape
        out = empty((S, k), np.float64)
        d_norm = asarray(np.ascontiguousarray(rxnormsq, dtype=np.float64))
        _lib.check(
            _lib.load().caf_sum_planes_qf2(ct.c_void_p(self.d_xcTemplates.ptr), self.numTemplates, S, k,
                                           sel.ctypes.data_as(ct.c_void_p), sel.size, ct.c_void_p(d_norm.ptr), ynormsq,
                                           ct.c_void_p(out.ptr), None),
            "caf_sum_planes_qf2",
        )
        return out

    def getCAF(self, templateIdx, numThreads=4):
        return self.getCAF_GPU(templateIdx).get()


def argmax2d(m):
    """ref: xcorrRoutines.py:815-830."""
    return np.unravel_index(np.argmax(m), m.shape)


def calcQF2(x, y):
    """ref: xcorrRoutines.py:833-848 (host helper for two aligned arrays)."""
    x = np.asarray(x)
    y = np.asarray(y)
    if x.ndim == 1 and y.ndim == 1:
        return np.abs(np.vdot(x, y)) ** 2 / np.linalg.norm(x) ** 2 / np.linalg.norm(y) ** 2
    if x.ndim == 2 and y.ndim == 2:
        xe = np.linalg.norm(x, axis=1) ** 2
        ye = np.linalg.norm(y, axis=1) ** 2
        return np.abs(np.sum(x * y.conj(), axis=1)) ** 2 / xe / ye


def convertQF2toSNR(qf2):
    """ref: xcorrRoutines.py:723-725."""
    return qf2 / (1.0 - qf2)


def convertQF2toEffSNR(qf2):
    """ref: xcorrRoutines.py:728-730."""
    return 2.0 * qf2 / (1.0 - qf2)


def convertEffSNRtoQF2(effSNR):
    """ref: xcorrRoutines.py:733-735."""
    return effSNR / (2 + effSNR)


def expectedEffSNR(snr1, snr2=np.inf, OSR=1):
    """ref: xcorrRoutines.py:738-755 (Stein)."""
    return 1.0 / (0.5 * (1 / snr1 + 1 / snr2 + 1 / snr1 / snr2)) / OSR


def sigmaDTO(signalBW, noiseBW, integTime, effSNR):
    """ref: xcorrRoutines.py:758-764."""
    return 1.0 / (np.pi / np.sqrt(3) * signalBW) / np.sqrt(noiseBW * integTime * effSNR)


def sigmaDFO(noiseBW, integTime, effSNR):
    """ref: xcorrRoutines.py:767-772."""
    return 0.55 / integTime / np.sqrt(noiseBW * integTime * effSNR)


def computeFastXcorrComplexity(N, K=1):
    """ref: xcorrRoutines.py:2084-2097."""
    return K * N * np.log2(N)


def computeGroupXcorrCZTcomplexity(m, L, n, K=1):
    """ref: xcorrRoutines.py:2100-2121."""
    Lc = next_fast_len(L + n)
    return K * m * 2 * Lc * np.log2(Lc)
