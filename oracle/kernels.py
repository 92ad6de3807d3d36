"""Oracle (test infrastructure): semantics of the reference's kernel-level ops.

Each function is a NumPy/SciPy statement of WHAT a reference CUDA kernel (and
its cupy launch wrapper) computes -- not of how the CUDA code is organised.
Tie-breaks follow NumPy / ippsMaxIndx (lowest index), not the CUDA thread-order
tie-break.  Not product code.

Reference locations:
  slidingMultiplyNormalised        custom_kernels/multiplySlices.cu:113-216, cupyExtensions.py:491-560
  multiTemplateSlidingDotProduct   custom_kernels/multiplySlices.cu:251-399, cupyExtensions.py:563-640
  multiplySlicesWithIndexedRows    custom_kernels/multiplySlices.cu:25-84,   cupyExtensions.py:405-488
  complex_magnSq_kernel            custom_kernels/complex_magn.cu:8-19,      cupyExtensions.py:337-387
  multiArgmaxAbsRows_complex64     custom_kernels/argmax.cu:93-153,          cupyExtensions.py:268-319
  movingAverage                    custom_kernels/filter.cu:291-347,         filterRoutines.py:1167-1203
  multiMovingAverage               custom_kernels/filter.cu:196-240,         filterRoutines.py:1129-1164
  movingComplexSum                 custom_kernels/filter.cu:374-438,         filterRoutines.py:1206-1238
  filter_smtaps*                   custom_kernels/filter.cu:9-181,           filterRoutines.py:417-575
  upfirdn_naive / upfirdn_sm       custom_kernels/upfirdn.cu:6-182,          filterRoutines.py:130-380
  findLocalMaxima                  custom_kernels/peakfinding.cu:14-58,      cupyExtensions.py:651-686
  copy*SlicesToMatrix / copyGroups custom_kernels/copying.cu:8-138,          cupyExtensions.py:17-215
"""

import numpy as np
import scipy.signal as sps
from numpy.lib.stride_tricks import sliding_window_view


def _pw64(x):
    x = np.asarray(x)
    return x.real.astype(np.float64) ** 2 + x.imag.astype(np.float64) ** 2


def slidingMultiplyNormalised(x, y, startIdx, idxlen, coefficient=None):
    """z[i, t] = x[t] * y[startIdx+i+t] / (||y[startIdx+i : +xlen]|| * coef), complex64.

    The template is NOT conjugated (caller's job, Appendix B.3); coef defaults to
    ||x||; window energy is accumulated in double then the divisor is cast to
    float32 (multiplySlices.cu:153,201-204); samples past the end of y read as 0.
    """
    x = np.asarray(x, dtype=np.complex64)
    y = np.asarray(y, dtype=np.complex64)
    n = x.size
    need = startIdx + idxlen + n - 1
    if need > y.size:
        y = np.concatenate((y, np.zeros(need - y.size, np.complex64)))
    if coefficient is None:
        coefficient = np.sqrt(np.sum(_pw64(x)))
    w = sliding_window_view(y, n)[startIdx : startIdx + idxlen]
    e = np.sum(_pw64(w), axis=1)
    div = (np.sqrt(e) * float(coefficient)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        return ((w * x) / div[:, None]).astype(np.complex64)


def multiTemplateSlidingDotProduct(x, templates, startIdx, idxlen, templateEnergies=None):
    """Per slide k: max over templates i of |sum_t T_i[t] x[k+t]|^2 / E_i / ||x[k:k+L]||^2.

    Returns (templateIdx int32, qf2 float32).  No conjugation in-kernel.  Strict
    '>' against an initial 0 => first (lowest) template index wins ties and an
    all-zero column reports (0, 0.0).
    """
    x = np.asarray(x, dtype=np.complex64)
    templates = np.asarray(templates, dtype=np.complex64)
    T, L = templates.shape
    if templateEnergies is None:
        templateEnergies = np.sum(_pw64(templates), axis=1).astype(np.float32)
    w = sliding_window_view(x, L)[startIdx : startIdx + idxlen]
    e = np.sum(_pw64(w), axis=1)
    dots = w.astype(np.complex128) @ templates.astype(np.complex128).T  # (idxlen, T)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = (np.abs(dots) ** 2 / np.asarray(templateEnergies, dtype=np.float64)[None, :] / e[:, None]).astype(np.float32)
    q = np.where(np.isnan(q), np.float32(0), q)
    ti = np.argmax(q, axis=1).astype(np.int32)
    return ti, q[np.arange(idxlen), ti]


def multiplySlicesWithIndexedRows(x, rows, sliceStarts, sliceLen, rowIdxs):
    """out[i, t] = rows[rowIdxs[i], t] * x[sliceStarts[i] + t]  (complex64)."""
    x = np.asarray(x, dtype=np.complex64)
    rows = np.asarray(rows, dtype=np.complex64)
    ss = np.asarray(sliceStarts, dtype=np.int64)
    g = x[ss[:, None] + np.arange(sliceLen)[None, :]]
    return (rows[np.asarray(rowIdxs), :sliceLen] * g).astype(np.complex64)


def complexMagnSq(x, out_dtype=np.float64):
    """re^2 + im^2 computed in the INPUT precision then cast (complex_magn.cu:17)."""
    x = np.asarray(x)
    return (x.real * x.real + x.imag * x.imag).astype(out_dtype)


def argmaxAbsRows(x, useNormSqInstead=False):
    """Row-wise (argmax uint32, max float32) of |x| or |x|^2; lowest index on ties;
    all-zero row -> (0, 0.0)."""
    x = np.asarray(x, dtype=np.complex64)
    v = (x.real * x.real + x.imag * x.imag).astype(np.float32)
    if not useNormSqInstead:
        v = np.abs(x).astype(np.float32)
    im = np.argmax(v, axis=1)
    return im.astype(np.uint32), v[np.arange(x.shape[0]), im]


def movingAverage(x, avgLength, sumInstead=False):
    """Causal moving mean/sum, same length as x, zeros assumed in front; double
    accumulation then float32 (== scipy.signal.lfilter(ones(L)[/L], 1, x))."""
    x = np.asarray(x, dtype=np.float32)
    cs = np.concatenate((np.zeros(avgLength, np.float64), np.cumsum(x.astype(np.float64))))
    s = cs[avgLength:] - cs[: x.size]
    return (s if sumInstead else s / float(avgLength)).astype(np.float32)


def multiMovingAverage(x, avgLength):
    """Row-wise causal moving mean of a 2-D float32 array."""
    x = np.asarray(x, dtype=np.float32)
    return np.stack([movingAverage(r, avgLength) for r in x])


def movingComplexSum(x, sumLength):
    """Valid-only forward moving complex sum, returned as |sum|^2 float32, length
    x.size - sumLength + 1 (== |np.convolve(x, ones(L), 'valid')|^2)."""
    x = np.asarray(x, dtype=np.complex64)
    cs = np.concatenate(([0.0], np.cumsum(x.astype(np.complex128))))
    s = cs[sumLength:] - cs[:-sumLength]
    return (s.real**2 + s.imag**2).astype(np.float32)


def filter_lfilter(x, taps, delay=None, dsr=1, dsPhase=0):
    """Causal FIR == scipy.signal.lfilter(taps, 1, x) with optional carried-in
    history ``delay`` (the samples preceding x) and decimation x[dsPhase::dsr]."""
    x = np.asarray(x)
    taps = np.asarray(taps, dtype=np.float32)
    if delay is not None and len(delay) > 0:
        ext = np.concatenate((np.asarray(delay, dtype=x.dtype), x))
        y = sps.lfilter(taps.astype(np.float64), 1, ext)[len(delay) :]
    else:
        y = sps.lfilter(taps.astype(np.float64), 1, x)
    return y[dsPhase::dsr].astype(x.dtype)


def upfirdn(x, taps, up, down, outabs=False):
    """== scipy.signal.upfirdn(taps, x, up, down); rows independently for 2-D x."""
    x = np.asarray(x)
    y = sps.upfirdn(np.asarray(taps, dtype=np.float64), x, up, down, axis=-1)
    if outabs:
        return np.abs(y).astype(np.float32)
    return y.astype(x.dtype)


def upfirdn_size(originalSize, tapsSize, up, down):
    """ref: filterRoutines.py:130-132."""
    return int(np.ceil((originalSize * up - (up - 1) + tapsSize - 1) / down))


def findLocalMaxima(x, minHeight):
    """Indices i with x[i] > minHeight and x[i] > x[i-1] and x[i] > x[i+1]
    (out-of-range neighbours read as 0).  The CUDA kernel's output order is
    nondeterministic (atomic compaction); the oracle returns ascending indices."""
    x = np.asarray(x, dtype=np.float32)
    p = np.concatenate(([0.0], x, [0.0])).astype(np.float32)
    m = (x > np.float32(minHeight)) & (x > p[:-2]) & (x > p[2:])
    return np.nonzero(m)[0].astype(np.int32)


def topk_peaks(x, minHeight, k):
    """Deterministic top-k local maxima: value descending, index ascending
    (SURVEY 7.2 step 7).  Returns indices int32 (<= k of them)."""
    idx = findLocalMaxima(x, minHeight)
    v = np.asarray(x, dtype=np.float32)[idx]
    order = np.lexsort((idx, -v))
    return idx[order][:k]


def copySlicesToMatrix(x, starts, length):
    """Row i = x[starts[i] : starts[i]+length] (complex64)."""
    x = np.asarray(x, dtype=np.complex64)
    s = np.asarray(starts, dtype=np.int64)
    return x[s[:, None] + np.arange(length)[None, :]]


def copyIncrementalEqualSlicesToMatrix(x, start, increment, length, rows):
    """Row i = x[start + i*increment : ... + length] (complex64)."""
    return copySlicesToMatrix(x, start + increment * np.arange(rows), length)


def dotTonesScaling(f0, fstep, numFreqs, src):
    """ref: genTones.cu:165-283 / spectralRoutines.py:580-630 (cupy-only upstream: restated from the kernel text,
    in float64 -- the kernel's own float recurrence over the frequencies is its error, not its definition).
    out[b, k] = sum over the 64-sample block b of src[i] * exp(j 2 pi (f0 + k fstep) i); complex128 (B, numFreqs).
    Pinned by the upstream docstring's identity sum(axis=0) == czt(src) at -(f0 + k fstep), checked against the
    importable reference `czt` in tests/golden/make_golden.py."""
    src = np.asarray(src).astype(np.complex128)
    n = src.size
    nb = (n + 63) // 64
    i = np.arange(nb * 64, dtype=np.float64)
    padded = np.zeros(nb * 64, np.complex128)
    padded[:n] = src
    out = np.empty((nb, numFreqs), np.complex128)
    for k in range(numFreqs):
        out[:, k] = (padded * np.exp(2j * np.pi * ((f0 + k * fstep) * i % 1.0))).reshape(nb, 64).sum(axis=1)
    return out


def copyGroups(x, y, xStarts, yStarts, lengths):
    """y[yStarts[b] + i] = x[xStarts[b] + i] for i < lengths[b]; returns y."""
    for xs, ys, l in zip(xStarts, yStarts, lengths):
        y.reshape(-1)[ys : ys + l] = x[xs : xs + l]
    return y
