"""FIR / upfirdn / moving-sum wrappers with the reference's names and argument meaning
(filterRoutines.py:95-575, 1129-1238), on ``DeviceArray``.

Semantics pinned by the reference itself: FIR == scipy.signal.lfilter(taps, 1, x)
(benchmark_filterkernels.py:72-74), upfirdn == scipy.signal.upfirdn
(benchmark_upfirdnkernels.py:58-67), moving average == lfilter(ones(L)/L)
(filterRoutines.py:1256), moving complex sum == |np.convolve(x, ones(L), 'valid')|^2 (:1358).
CUDA launch-tuning kwargs are accepted and ignored.
"""

import ctypes as ct

import numpy as np

from . import _lib
from .devarray import DeviceArray, asarray, empty, requireDtype, zeros


def _p(a):
    return ct.c_void_p(a.ptr) if a is not None else None


class CupyKernelFilter:
    """ref: filterRoutines.py:95-575."""

    def __init__(self, memory=None, memory_dtype=np.complex64):
        self.delay = zeros(memory, memory_dtype) if memory is not None else None

    def resetDelay(self):
        if self.delay is not None:
            self.delay = zeros(self.delay.size, self.delay.dtype)

    @staticmethod
    def getUpfirdnSize(originalSize, tapsSize, up, down):
        """ref: filterRoutines.py:130-132 (matches scipy.signal.upfirdn's output length)."""
        return int(np.ceil((originalSize * up - (up - 1) + tapsSize - 1) / down))

    # -- upfirdn -------------------------------------------------------------------------
    def upfirdn_sm(self, d_x, d_taps, up, down, THREADS_PER_BLOCK=256, alsoReturnAbs=False, d_out=None, d_outabs=None):
        """Row-wise upfirdn of a 2-D complex64 matrix (ref: filterRoutines.py:134-258, upfirdn.cu:68-182)."""
        requireDtype(np.complex64, d_x)
        requireDtype(np.float32, d_taps)
        if d_x.ndim != 2:
            raise ValueError("d_x must be 2D.")
        rows, n = d_x.shape
        outlen = self.getUpfirdnSize(n, d_taps.size, up, down)
        if d_out is None:
            d_out = empty((rows, outlen), np.complex64)
        else:
            if d_out.shape != (rows, outlen):
                raise ValueError("d_out must have dimensions (%d, %d)." % (rows, outlen))
            requireDtype(np.complex64, d_out)
        if alsoReturnAbs:
            if d_outabs is None:
                d_outabs = empty((rows, outlen), np.float32)
            else:
                if d_outabs.shape != (rows, outlen):
                    raise ValueError("d_outabs must have dimensions (%d, %d)." % (rows, outlen))
                requireDtype(np.float32, d_outabs)
        else:
            d_outabs = None
        _lib.check(_lib.load().caf_upfirdn(_p(d_x), rows, n, _p(d_taps), d_taps.size, int(up), int(down), _p(d_out),
                                           _p(d_outabs), outlen, None))
        return (d_out, d_outabs) if alsoReturnAbs else d_out

    def upfirdn_naive(self, d_x, d_taps, up, down, THREADS_PER_BLOCK=256, alsoReturnAbs=False, d_out=None,
                      d_outabs=None):
        """1-D upfirdn (ref: filterRoutines.py:260-380, upfirdn.cu:6-59)."""
        if d_x.dtype != np.complex64:
            raise TypeError("d_x must be complex64.")
        if d_taps.dtype != np.float32:
            raise TypeError("d_taps must be float32.")
        outlen = self.getUpfirdnSize(d_x.size, d_taps.size, up, down)
        if d_out is None:
            d_out = empty(outlen, np.complex64)
        else:
            if d_out.dtype != np.complex64:
                raise TypeError("d_out must be complex64.")
            if d_out.size < outlen:
                raise ValueError("d_out must be at least length %d" % outlen)
        if alsoReturnAbs:
            if d_outabs is None:
                d_outabs = empty(outlen, np.float32)
            else:
                if d_outabs.dtype != np.float32:
                    raise TypeError("d_outabs must be float32.")
                if d_outabs.size < outlen:
                    raise ValueError("d_outabs must be at least length %d" % outlen)
        else:
            d_outabs = None
        _lib.check(_lib.load().caf_upfirdn(_p(d_x), 1, d_x.size, _p(d_taps), d_taps.size, int(up), int(down), _p(d_out),
                                           _p(d_outabs), outlen, None))
        return (d_out, d_outabs) if alsoReturnAbs else d_out

    def run_upfirdn(self, d_x, d_taps, up, down, THREADS_PER_BLOCK=256):
        """Streaming upfirdn with carried-in history (ref: filterRoutines.py:382-415)."""
        if self.delay is None:
            raise TypeError("Delay has not been allocated. Re-initialize with memory argument.")
        dl = self.delay.size
        d_xext = empty(dl + d_x.size, np.complex64)
        lib = _lib.load()
        _lib.check(lib.caf_d2d(ct.c_void_p(d_xext.ptr), _p(self.delay), dl * 8, None))
        _lib.check(lib.caf_d2d(ct.c_void_p(d_xext.ptr + dl * 8), _p(d_x), d_x.size * 8, None))
        d_out = self.upfirdn_naive(d_xext, d_taps, up, down)
        _lib.check(lib.caf_d2d(_p(self.delay), ct.c_void_p(d_x.ptr + (d_x.size - dl) * 8), dl * 8, None))
        _lib.check(lib.caf_stream_sync(None))
        length2return = int(d_x.size * up // down)
        skip = int(dl * up // down)
        return d_out[skip : skip + length2return]

    # -- FIR -----------------------------------------------------------------------------
    def filter_smtaps(self, d_x, d_taps, THREADS_PER_BLOCK=128, OUTPUT_PER_BLK=128, useInternalDelay=False, dsr=1,
                      dsPhase=0):
        """lfilter(taps, 1, x)[dsPhase::dsr] (ref: filterRoutines.py:417-480, filter.cu:9-58)."""
        requireDtype(np.float32, d_taps)
        requireDtype(np.complex64, d_x)
        assert d_x.ndim == 1 and d_taps.ndim == 1
        if dsPhase >= dsr or dsPhase < 0:
            raise ValueError("dsPhase must be between in the range [0,dsr-1].")
        outlength = (d_x.size - dsPhase) // dsr + (1 if (d_x.size - dsPhase) % dsr else 0)
        d_out = empty(outlength, np.complex64)
        delay = self.delay if useInternalDelay else None
        if useInternalDelay and delay is None:
            raise TypeError("Delay has not been allocated. Re-initialize with memory argument.")
        # (no tap limit: beyond a few hundred taps per kept output libcaf switches to its overlap-save form, any length;
        # the reference's kernel stops where the taps no longer fit its shared memory, cupyHelpers.py:50-77)
        _lib.check(_lib.load().caf_fir_lfilter(_p(d_x), d_x.size, _p(d_taps), d_taps.size, _p(delay),
                                               delay.size if delay is not None else 0, int(dsr), int(dsPhase),
                                               _p(d_out), outlength, None))
        return d_out

    def run_filter_smtaps(self, d_x, d_taps, THREADS_PER_BLOCK=128, OUTPUT_PER_BLK=128):
        """Streaming FIR: uses and then updates the carried-in history (ref: filterRoutines.py:482-501)."""
        if self.delay is None:
            raise TypeError("Delay has not been allocated. Re-initialize with memory argument.")
        d_out = self.filter_smtaps(d_x, d_taps, useInternalDelay=True)
        dl = self.delay.size
        lib = _lib.load()
        _lib.check(lib.caf_d2d(_p(self.delay), ct.c_void_p(d_x.ptr + (d_x.size - dl) * 8), dl * 8, None))
        _lib.check(lib.caf_stream_sync(None))
        return d_out

    def filter_smtaps_sminput(self, d_x, d_taps, THREADS_PER_BLOCK=128, OUTPUT_PER_BLK=256):
        """ref: filterRoutines.py:503-575 (filter.cu:61-181).  complex64 or float32 input."""
        requireDtype(np.float32, d_taps)
        requireDtype([np.complex64, np.float32], d_x)
        if d_x.ndim != 1:
            raise ValueError("d_x must be 1D.")
        if d_taps.ndim != 1:
            raise ValueError("d_taps must be 1D.")
        if d_x.dtype == np.float32:
            # real input: run the complex kernel on (x + 0j); the imaginary lane stays exactly zero
            xc = asarray(d_x.get().astype(np.complex64))
            yc = self.filter_smtaps(xc, d_taps)
            return asarray(np.ascontiguousarray(yc.get().real))
        return self.filter_smtaps(d_x, d_taps)


def cupyMultiMovingAverage(d_x, avgLength, THREADS_PER_BLOCK=32):
    """Row-wise causal moving mean (ref: filterRoutines.py:1129-1164, filter.cu:196-240)."""
    requireDtype(np.float32, d_x)
    if d_x.ndim == 1:
        raise ValueError("Input array must be 2D. For single arrays just use cupy's filter functions.")
    if d_x.shape[0] == 0:
        raise ValueError("Input array must have at least one row.")
    d_avg = empty(d_x.shape, np.float32)
    _lib.check(_lib.load().caf_moving_average(_p(d_x), d_x.shape[0], d_x.shape[1], int(avgLength), 0, _p(d_avg), None))
    return d_avg


def cupyMovingAverage(x, avgLength, NUM_PER_THREAD=33, THREADS_PER_BLK=32, sumInstead=False):
    """Causal moving mean / sum, same length as the input (ref: filterRoutines.py:1167-1203,
    filter.cu:291-347).  On the CAF path: sliding rx energy for QF (xcorrRoutines.py:342-348)."""
    requireDtype(np.float32, x)
    if NUM_PER_THREAD % 2 == 0:
        raise ValueError("NUM_PER_THREAD must be odd.")
    d_out = empty(x.shape, np.float32)
    _lib.check(_lib.load().caf_moving_average(_p(x), 1, x.size, int(avgLength), 1 if sumInstead else 0, _p(d_out), None))
    return d_out


def cupyComplexMovingSum(x, sumLength, NUM_PER_THREAD=33, THREADS_PER_BLK=32, sumInstead=True):
    """|valid-only moving complex sum|^2, float32 (ref: filterRoutines.py:1206-1238, filter.cu:374-438)."""
    requireDtype(np.complex64, x)
    if NUM_PER_THREAD % 2 == 0:
        raise ValueError("NUM_PER_THREAD must be odd.")
    d_out = empty(x.size - sumLength + 1, np.float32)
    _lib.check(_lib.load().caf_complex_moving_sum(_p(x), x.size, int(sumLength), _p(d_out), None))
    return d_out


__all__ = ["CupyKernelFilter", "cupyMultiMovingAverage", "cupyMovingAverage", "cupyComplexMovingSum", "DeviceArray"]
