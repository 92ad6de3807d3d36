"""CAF + chirp-Z fine zoom around the top-k peaks (BASELINE.json config 5).

Replaces the reference's two-stage workflow -- a coarse xcorr, then ``cztXcorr`` /
``pbIppGroupXcorrCZT`` over a narrow frequency span at the delays of interest
(xcorrRoutines.py:413-457, benchmarks/benchmark_czts.py:31-82,
benchmarks/benchmark_groupXcorrs.py:37-72) -- with ONE libcaf call, ``caf_zoom_czt`` (include/caf.h):

  coarse per-delay trace of a finished CAFPlan.run -> local maxima (peakfinding.cu:52 predicate) -> top-k on the
  device (value descending, delay ascending) -> all normalised product rows in one launch -> one batched
  Bluestein CZT on the fine grid -> |.|^2 -> refined (frequency, QF^2).

This module is a thin caller: it allocates the k-row result table, makes the call and reads the table back.
"""

import ctypes as ct

import numpy as np

from . import _lib
from .devarray import DeviceArray, asarray, empty, requireDeviceArray


def zoom_num_bins(span, step):
    """Number of fine-grid bins of a zoom over -span ... +span in steps of ``step`` (any common unit)."""
    nb = ct.c_int32()
    _lib.check(_lib.load().caf_zoom_num_bins(float(span), float(step), ct.byref(nb)), "caf_zoom_num_bins")
    return nb.value


def zoom_czt(plan, d_rx, plan_result, k=8, min_height=0.0, span=None, step=None, template=0, shift_start=0,
             planes=False):
    """``caf_zoom_czt`` on the trace of ``template`` in ``plan_result`` (a CAFPlan.run(..., rows=True) result of
    ``plan`` over ``d_rx``).  ``span`` / ``step`` are in cycles per sample.  Returns a dict of host arrays with one
    entry per peak found, best first: delay, coarse_index, coarse_qf2, fine_index, fine_freq (cycles per sample),
    fine_qf2 (+ planes (npeaks, nbins) float32 if requested)."""
    requireDeviceArray(d_rx)
    if plan_result.row_max is None or plan_result.row_arg is None:
        raise ValueError("the coarse result needs per-delay rows (run the plan with rows=True)")
    T, S = plan_result.row_max.shape
    if not (0 <= template < T):
        raise ValueError("template index outside the coarse result")
    k = int(k)
    nb = zoom_num_bins(span, step)
    cnt = empty(1, np.int32)
    dly, cidx, fidx = empty(k, np.int32), empty(k, np.int32), empty(k, np.int32)
    cq, fq = empty(k, np.float32), empty(k, np.float32)
    ff = empty(k, np.float64)
    pl = empty((k, nb), np.float32) if planes else None
    o = _lib.CafZoomOutputs(cnt.ptr, dly.ptr, cidx.ptr, cq.ptr, fidx.ptr, ff.ptr, fq.ptr, pl.ptr if planes else None)
    _lib.check(
        _lib.load().caf_zoom_czt(plan._h, int(template), ct.c_void_p(d_rx.ptr), d_rx.size,
                                 ct.c_void_p(plan_result.row_max.ptr + 4 * S * template),
                                 ct.c_void_p(plan_result.row_arg.ptr + 4 * S * template), int(shift_start), S, k,
                                 float(min_height), float(span), float(step), ct.byref(o), None),
        "caf_zoom_czt",
    )
    n = int(cnt.get()[0])
    if n < 0:
        raise ValueError("more than 2^20 local maxima above min_height=%g; raise min_height" % min_height)
    res = {"delay": dly.get()[:n].astype(np.int64), "coarse_index": cidx.get()[:n], "coarse_qf2": cq.get()[:n],
           "fine_index": fidx.get()[:n], "fine_freq": ff.get()[:n], "fine_qf2": fq.get()[:n]}
    if planes:
        res["planes"] = pl.get()[:n]
    return res


def caf_with_zoom(plan, d_rx, plan_result, bins, grid, fs, k=8, min_height=None, span_bins=1.0, step_bins=1.0 / 64,
                  template=0):
    """Config-5 pipeline on an existing coarse result (``plan.run(d_rx, rows=True)``; on-grid bins ``bins``/``grid``).

    Returns a list of dicts (delay, coarse_bin, coarse_qf2, fine_freq [Hz], fine_qf2), best first."""
    if min_height is None:
        min_height = 0.25 * float(plan_result.peak_val.get()[template]) if plan_result.peak_val is not None else 0.0
    r = zoom_czt(plan, d_rx, plan_result, k=k, min_height=min_height, span=span_bins / float(grid),
                 step=step_bins / float(grid), template=template)
    cb = np.asarray(bins)[r["coarse_index"]]
    return [
        {"delay": int(d), "coarse_bin": int(b), "coarse_qf2": float(v), "fine_freq": float(f) * fs, "fine_qf2": float(q)}
        for d, b, v, f, q in zip(r["delay"], cb, r["coarse_qf2"], r["fine_freq"], r["fine_qf2"])
    ]


# ------------------------------------------------------------------------------------------------------------------
# The same pipeline spelled out with the kernel-level wrappers (local maxima -> host ordering of the candidates ->
# one product row per peak -> batched CZT).  NOT used by caf_with_zoom any more; kept because the tests cross-check
# caf_zoom_czt's device top-k and batched rows against this independent chain.
# ------------------------------------------------------------------------------------------------------------------
from .cupyExtensions import (  # noqa: E402
    cupyArgmaxAbsRows_complex64,
    cupyComplexMagnSq,
    cupyFindLocalMaxima,
    multiplySlidesNormalised,
)
from .spectralRoutines import CZTCachedGPU  # noqa: E402


def gather(d_x, d_idx, n=None):
    """d_x[d_idx[:n]] on the device for float32 / int32 arrays (``caf_gather_b32``); returns a host array."""
    requireDeviceArray(d_x)
    if d_x.dtype.itemsize != 4:
        raise TypeError("4-byte element types only, found %s" % d_x.dtype)
    if d_idx.dtype != np.int32:
        raise TypeError("Must be int32, found %s" % d_idx.dtype)
    n = d_idx.size if n is None else int(n)
    out = empty(max(n, 1), d_x.dtype)
    _lib.check(_lib.load().caf_gather_b32(ct.c_void_p(d_x.ptr), d_x.size, ct.c_void_p(d_idx.ptr), n, ct.c_void_p(out.ptr),
                                          None), "caf_gather_b32")
    return out.get()[:n]


def topk_local_maxima(d_trace, k, min_height, maxNumPeaks=100000):
    """Indices of the k largest local maxima of a float32 device trace: value descending, index ascending."""
    requireDeviceArray(d_trace)
    idx, cnt = cupyFindLocalMaxima(d_trace, float(min_height), maxNumPeaks=maxNumPeaks)
    n = int(cnt.get()[0])
    if n > maxNumPeaks:
        raise ValueError("%d local maxima above min_height=%g exceed maxNumPeaks=%d; raise min_height" % (n, min_height, maxNumPeaks))
    if n == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.float32)
    # only the candidates leave the device: their values are gathered there (a 2^24-sample trace is 67 MB)
    vals = gather(d_trace, idx, n)
    idx = idx.get()[:n]
    order = np.lexsort((idx, -vals))[:k]
    return idx[order].astype(np.int64), vals[order]


_ZOOM_CZT = {}  # (n, span, step, fs) -> CZTCachedGPU on the relative grid -span ... +span


def czt_zoom(cutout, d_rx, delays, coarse_freqs, fs, span, step):
    """For each (delay, coarse frequency) evaluate QF^2 on the fine grid coarse-span ... coarse+span (Hz) in
    steps of ``step`` and return (fine_freq float64[k], qf2 float32[k], planes list of float32 arrays).

    All peaks share ONE batched CZT: the product row of peak i is rotated by exp(-j 2 pi f0_i n / fs), which moves
    its grid f0_i - span ... f0_i + span onto the common relative grid -span ... +span (cached chirp constants)."""
    requireDeviceArray(d_rx)
    cutout = np.ascontiguousarray(cutout, dtype=np.complex64)
    n = cutout.size
    npk = len(delays)
    if npk == 0:
        return np.zeros(0, np.float64), np.zeros(0, np.float32), []
    d_cc = asarray(cutout.conj())
    lib = _lib.load()
    rows = empty((npk, n), np.complex64)
    for i, d in enumerate(delays):
        # (1, n): rx[d:d+n] * conj(cutout) / norms -- on the n-sample view, so that the energy pass covers n samples
        row = multiplySlidesNormalised(d_cc, d_rx[int(d) : int(d) + n], 0, 1)
        _lib.check(lib.caf_d2d(ct.c_void_p(rows.ptr + 8 * n * i), ct.c_void_p(row.ptr), 8 * n, None), "caf_d2d")
    f0 = np.asarray(coarse_freqs, dtype=np.float64)
    cyc = np.outer(f0 / fs, np.arange(n, dtype=np.float64))
    rot = np.exp(2j * np.pi * (cyc - np.floor(cyc))).astype(np.complex64)  # rows * conj(rot)
    d_rot = asarray(rot)
    _lib.check(lib.caf_mul_conj(ct.c_void_p(rows.ptr), ct.c_void_p(d_rot.ptr), npk * n, ct.c_void_p(rows.ptr), None),
               "caf_mul_conj")
    key = (n, float(span), float(step), float(fs))
    cz = _ZOOM_CZT.get(key)
    if cz is None:
        if len(_ZOOM_CZT) > 16:
            _ZOOM_CZT.clear()
        cz = _ZOOM_CZT[key] = CZTCachedGPU(n, -span, span, step, fs)
    z = cz.runMany(rows)  # (npk, k) complex64 on the device
    am, mx = cupyArgmaxAbsRows_complex64(z, returnMaxValues=True, useNormSqInstead=True)
    j = am.get().astype(np.int64)
    # the labels of a per-peak CZT object: f1 + j * step with f1 = f0 - span (CZTCached.getFreq)
    fine_f = j * step + (f0 - span)
    fine_q = mx.get().astype(np.float32)
    pl = cupyComplexMagnSq(z, np.float32).get()
    return fine_f, fine_q, [pl[i] for i in range(npk)]


__all__ = ["zoom_num_bins", "zoom_czt", "caf_with_zoom", "gather", "topk_local_maxima", "czt_zoom", "DeviceArray"]
