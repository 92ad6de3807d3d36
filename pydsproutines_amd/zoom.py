"""CAF + chirp-Z fine zoom around the top-k peaks (BASELINE.json config 5).

Replaces the reference's two-stage workflow -- a coarse xcorr, then ``cztXcorr`` /
``pbIppGroupXcorrCZT`` over a narrow frequency span at the delays of interest
(xcorrRoutines.py:413-457, benchmarks/benchmark_czts.py:31-82,
benchmarks/benchmark_groupXcorrs.py:37-72) -- with device kernels end to end:

  coarse per-delay trace (CAFPlan) -> local maxima (peakfinding.cu:52 predicate, ``caf_find_local_maxima``)
  -> top-k (value descending, index ascending) -> normalised product row per peak
  (``caf_sliding_multiply_normalised``) -> Bluestein CZT on a fine grid (``caf_czt_run_many``)
  -> |.|^2 -> refined (frequency, QF^2).

Only the tiny peak list (<= maxNumPeaks indices) is ordered on the host.
"""

import numpy as np

from .cupyExtensions import (
    cupyArgmaxAbsRows_complex64,
    cupyComplexMagnSq,
    cupyFindLocalMaxima,
    multiplySlidesNormalised,
)
from .devarray import DeviceArray, asarray, requireDeviceArray
from .spectralRoutines import CZTCachedGPU


def topk_local_maxima(d_trace, k, min_height, maxNumPeaks=100000):
    """Indices of the k largest local maxima of a float32 device trace: value descending, index ascending."""
    requireDeviceArray(d_trace)
    idx, cnt = cupyFindLocalMaxima(d_trace, float(min_height), maxNumPeaks=maxNumPeaks)
    n = int(cnt.get()[0])
    if n > maxNumPeaks:
        raise ValueError("%d local maxima above min_height=%g exceed maxNumPeaks=%d; raise min_height" % (n, min_height, maxNumPeaks))
    idx = idx.get()[:n]
    if n == 0:
        return idx.astype(np.int64), np.zeros(0, np.float32)
    # fetch only the candidate values (one contiguous read of the trace is cheaper than n tiny copies)
    vals = d_trace.get()[idx]
    order = np.lexsort((idx, -vals))[:k]
    return idx[order].astype(np.int64), vals[order]


def czt_zoom(cutout, d_rx, delays, coarse_freqs, fs, span, step):
    """For each (delay, coarse frequency) evaluate QF^2 on the fine grid coarse-span ... coarse+span (Hz) in
    steps of ``step`` and return (fine_freq float64[k], qf2 float32[k], planes list of float32 arrays)."""
    requireDeviceArray(d_rx)
    cutout = np.ascontiguousarray(cutout, dtype=np.complex64)
    n = cutout.size
    d_cc = asarray(cutout.conj())
    fine_f = np.zeros(len(delays), np.float64)
    fine_q = np.zeros(len(delays), np.float32)
    planes = []
    for i, (d, f0) in enumerate(zip(delays, coarse_freqs)):
        row = multiplySlidesNormalised(d_cc, d_rx, int(d), 1)  # (1, n): rx[d:d+n] * conj(cutout) / norms
        cz = CZTCachedGPU(n, f0 - span, f0 + span, step, fs)
        z = cz.runMany(row)  # (1, k) complex64 on the device
        am, mx = cupyArgmaxAbsRows_complex64(z, returnMaxValues=True, useNormSqInstead=True)
        j = int(am.get()[0])
        fine_f[i] = cz.getFreq()[j]
        fine_q[i] = mx.get()[0]
        planes.append(cupyComplexMagnSq(z, np.float32).get()[0])
    return fine_f, fine_q, planes


def caf_with_zoom(cutout, d_rx, plan_result, bins, grid, fs, k=8, min_height=None, span_bins=1.0, step_bins=1.0 / 64):
    """Config-5 pipeline on an existing coarse result (``CAFPlan.run(..., rows=True)`` of ONE template).

    Returns a list of dicts (delay, coarse_bin, coarse_qf2, fine_freq, fine_qf2), best first."""
    trace = plan_result.row_max[0]
    args = plan_result.row_arg[0]
    if min_height is None:
        min_height = 0.25 * float(plan_result.peak_val.get()[0]) if plan_result.peak_val is not None else 0.0
    delays, vals = topk_local_maxima(trace, k, min_height)
    if delays.size == 0:
        return []
    a = args.get()
    cbins = np.asarray(bins)[a[delays]]
    bin_hz = fs / float(grid)
    ff, fq, _ = czt_zoom(cutout, d_rx, delays, cbins * bin_hz, fs, span_bins * bin_hz, step_bins * bin_hz)
    return [
        {"delay": int(d), "coarse_bin": int(b), "coarse_qf2": float(v), "fine_freq": float(f), "fine_qf2": float(q)}
        for d, b, v, f, q in zip(delays, cbins, vals, ff, fq)
    ]


__all__ = ["topk_local_maxima", "czt_zoom", "caf_with_zoom", "DeviceArray"]
