"""Kernel-level wrappers with the reference's names, argument meaning and error behaviour
(cupyExtensions.py, cupyHelpers.py), operating on ``DeviceArray`` instead of ``cupy.ndarray``.

CUDA tuning kwargs (THREADS_PER_BLOCK, numSlidesPerBlk, NUM_BLKS ...) are accepted and ignored:
the CDNA4 kernels pick their own tiling and have no 48 000-byte shared-memory ceiling
(SURVEY Appendix B.11), so the MemoryError of ``cupyCheckExceedsSharedMem`` is only raised where a
real limit of this implementation is hit.  dtype checks raise TypeError, shape/range checks
ValueError, exactly as upstream.  Argmax tie-breaks follow NumPy (lowest index).
"""

import ctypes as ct

import numpy as np

from . import _lib
from .devarray import (  # noqa: F401
    DeviceArray,
    asarray,
    asnumpy,
    empty,
    requireDeviceArray,
    requireDtype,
    zeros,
)

requireCupyArray = requireDeviceArray  # upstream name (cupyHelpers.py:74-77)
cupyRequireDtype = requireDtype  # upstream name (cupyHelpers.py:50-69)


def _p(a):
    return ct.c_void_p(a.ptr) if a is not None else None


def cupyCheckExceedsSharedMem(requestedBytes, maximumBytes=160 * 1024):
    """ref: cupyHelpers.py:72-77; the limit here is the 160 KiB LDS of a CDNA4 CU."""
    if requestedBytes > maximumBytes:
        raise MemoryError("Shared memory requested %d bytes exceeds maximum %d bytes" % (requestedBytes, maximumBytes))


def cupyGetEnoughBlocks(length, computedPerBlock):
    """ref: cupyHelpers.py:80-88."""
    return length // computedPerBlock + (1 if length % computedPerBlock else 0)


# ---- copies -------------------------------------------------------------------------------
def cupyCopyGroups32fc(x, y, xStarts, yStarts, lengths, threads_per_blk=256):
    """ref: cupyExtensions.py:41-83.  y[yStarts[b]+i] = x[xStarts[b]+i], i < lengths[b]."""
    assert len(xStarts) == len(yStarts) and len(xStarts) == len(lengths)
    _lib.check(_lib.load().caf_copy_groups(_p(x), _p(y), _p(xStarts), _p(yStarts), _p(lengths), len(xStarts), None))


def cupyCopySlicesToMatrix_32fc(d_x, d_sliceBounds, rowLength=None, THREADS_PER_BLOCK=128):
    """ref: cupyExtensions.py:101-127, copying.cu:8-38.  Rows = x[start:end] for each (start, end)."""
    requireDtype(np.complex64, d_x)
    requireDtype(np.int32, d_sliceBounds)
    numSlices = d_sliceBounds.shape[0]
    if rowLength is None:
        b = d_sliceBounds.get()
        rowLength = int(np.max(b[:, 1] - b[:, 0]))
    d_out = empty((numSlices, rowLength), np.complex64)
    _lib.check(_lib.load().caf_copy_slices_to_matrix(_p(d_x), d_x.size, _p(d_sliceBounds), 2, 0, 0, int(rowLength),
                                                     numSlices, _p(d_out), None))
    return d_out


def cupyCopyEqualSlicesToMatrix_32fc(d_x, d_xStartIdxs, rowLength, d_out=None):
    """ref: cupyExtensions.py:130-156, copying.cu:40-66."""
    requireDtype(np.complex64, d_x)
    requireDtype(np.int32, d_xStartIdxs)
    if d_out is None:
        d_out = empty((d_xStartIdxs.size, rowLength), np.complex64)
    else:
        requireDtype(np.complex64, d_out)
        if d_out.shape != (d_xStartIdxs.size, rowLength):
            raise ValueError("d_out must have the shape %d, %d" % (d_xStartIdxs.size, rowLength))
    _lib.check(_lib.load().caf_copy_slices_to_matrix(_p(d_x), d_x.size, _p(d_xStartIdxs), 1, 0, 0, int(rowLength),
                                                     d_xStartIdxs.size, _p(d_out), None))
    return d_out


def cupyCopyIncrementalEqualSlicesToMatrix_32fc(d_x, startIdx, increment, rowLength, numRows, d_out=None):
    """ref: cupyExtensions.py:159-215, copying.cu:86-138."""
    requireDtype(np.complex64, d_x)
    if d_out is None:
        d_out = empty((numRows, rowLength), np.complex64)
    else:
        requireDtype(np.complex64, d_out)
        if d_out.shape != (numRows, rowLength):
            raise ValueError("d_out must have the shape %d, %d" % (numRows, rowLength))
    _lib.check(_lib.load().caf_copy_slices_to_matrix(_p(d_x), d_x.size, None, 1, int(startIdx), int(increment),
                                                     int(rowLength), int(numRows), _p(d_out), None))
    return d_out


# ---- argmax / magnitude -------------------------------------------------------------------
def cupyArgmax3d_uint32(d_x, THREADS_PER_BLOCK=128, alsoReturnMaxValue=False):
    """ref: cupyExtensions.py:225-265, argmax.cu:11-81.  Argmax over the last 3 dims of a 4-D uint32 array."""
    if d_x.dtype != np.uint32:
        raise TypeError("d_x must be uint32.")
    if d_x.ndim != 4:
        raise ValueError("d_x must be 4-d. Argmax taken over the last 3 dimensions.")
    numItems, dim1, dim2, dim3 = d_x.shape
    d_argmax = empty((numItems, 3), np.uint32)
    d_max = empty(numItems, np.uint32) if alsoReturnMaxValue else None
    _lib.check(_lib.load().caf_argmax3d_u32(_p(d_x), numItems, dim1, dim2, dim3, _p(d_argmax), _p(d_max), None))
    if alsoReturnMaxValue:
        return d_argmax, d_max
    return d_argmax



def cupyArgmaxAbsRows_complex64(d_x, d_argmax=None, d_max=None, returnMaxValues=False, THREADS_PER_BLOCK=128,
                                useNormSqInstead=False):
    """ref: cupyExtensions.py:268-319, argmax.cu:93-153.  Row-wise argmax of |x| (or |x|^2)."""
    requireDtype(np.complex64, d_x)
    numRows, length = d_x.shape
    if d_argmax is None:
        d_argmax = empty(numRows, np.uint32)
    else:
        requireDtype(np.uint32, d_argmax)
        if d_argmax.shape != (numRows,):
            raise ValueError("d_argmax shape must be 1D of length %d" % numRows)
    if returnMaxValues:
        if d_max is None:
            d_max = empty(numRows, np.float32)
        else:
            requireDtype(np.float32, d_max)
            if d_max.shape != (numRows,):
                raise ValueError("d_max shape must be 1D of length %d" % numRows)
    else:
        d_max = None
    _lib.check(_lib.load().caf_argmax_abs_rows(_p(d_x), numRows, length, _p(d_argmax), _p(d_max),
                                               1 if useNormSqInstead else 0, None))
    if returnMaxValues:
        return d_argmax, d_max
    return d_argmax


def cupyComplexMagnSq(d_x, out_dtype=np.float64, THREADS_PER_BLOCK=128):
    """ref: cupyExtensions.py:337-387, complex_magn.cu:8-19.  c64->f32, c64->f64, c128->f64."""
    out_dtype = np.dtype(out_dtype)
    if out_dtype == np.float32:
        requireDtype(np.complex64, d_x)
        out = empty(d_x.shape, np.float32)
        _lib.check(_lib.load().caf_complex_magnsq(_p(d_x), d_x.size, 0, _p(out), 0, None))
        return out
    if out_dtype == np.float64:
        if d_x.dtype not in (np.dtype(np.complex64), np.dtype(np.complex128)):
            raise TypeError("d_x must be complex64 or complex128.")
        out = empty(d_x.shape, np.float64)
        _lib.check(_lib.load().caf_complex_magnsq(_p(d_x), d_x.size, 1 if d_x.dtype == np.complex128 else 0, _p(out), 1,
                                                  None))
        return out
    raise TypeError("out_dtype must be float32 or float64.")


# ---- sliding products ---------------------------------------------------------------------
def multiplySlicesOptimistically(d_x, d_rows, d_sliceStarts, d_sliceLengths, d_rowIdxs, THREADS_PER_BLOCK=256,
                                 NUM_BLKS=None, outlength=None):
    """ref: cupyExtensions.py:405-488, multiplySlices.cu:25-84."""
    if d_x.dtype != np.complex64 or d_rows.dtype != np.complex64:
        raise TypeError("Inputs x and rows must be complex64.")
    if d_sliceStarts.dtype != np.int32 or d_sliceLengths.dtype != np.int32 or d_rowIdxs.dtype != np.int32:
        raise TypeError("sliceStarts, sliceLengths & rowIdxs must be int32.")
    if d_x.ndim != 1:
        raise ValueError("x should be 1-dimensional.")
    if d_rows.ndim != 2:
        raise ValueError("rows should be 2-dimensional.")
    numSlices = d_sliceStarts.size
    if d_sliceLengths.size != numSlices or d_rowIdxs.size != numSlices:
        raise ValueError("sliceLengths and rowIdxs should be same length as sliceStarts.")
    numRows, rowLength = d_rows.shape
    if outlength is None:
        outlength = rowLength
    if not np.all(d_sliceLengths.get() <= outlength):
        raise ValueError("Some slices exceed the output length!")
    d_out = empty((numSlices, outlength), np.complex64)
    _lib.check(_lib.load().caf_multiply_slices_indexed_rows(_p(d_x), d_x.size, _p(d_rows), numRows, rowLength,
                                                            _p(d_sliceStarts), _p(d_sliceLengths), _p(d_rowIdxs),
                                                            int(outlength), numSlices, _p(d_out), None))
    return d_out


def multiplySlidesNormalised(d_x, d_y, startIdx, idxlen, THREADS_PER_BLOCK=128, numSlidesPerBlk=None, coefficient=None):
    """ref: cupyExtensions.py:491-560, multiplySlices.cu:113-216.
    z[i, t] = x[t] * y[startIdx+i+t] / (||y window|| * coefficient); x is NOT conjugated here."""
    requireDtype(np.complex64, d_x)
    requireDtype(np.complex64, d_y)
    if startIdx < 0 or startIdx + idxlen > d_y.size:
        raise ValueError("startIdx and idxlen should be within the bounds of d_y.")
    if coefficient is None:
        coef = float(np.linalg.norm(d_x.get().astype(np.complex128)))
    else:
        if isinstance(coefficient, DeviceArray):
            requireDtype(np.float64, coefficient)
            if coefficient.size != 1:
                raise ValueError("coefficient should be a single element 1D array.")
            coef = float(coefficient.get().reshape(-1)[0])
        else:
            c = np.asarray(coefficient)
            if c.dtype != np.float64:
                raise TypeError("Must be float64, found %s" % c.dtype)
            if c.size != 1:
                raise ValueError("coefficient should be a single element 1D array.")
            coef = float(c.reshape(-1)[0])
    d_pdts = empty((idxlen, d_x.size), np.complex64)
    _lib.check(_lib.load().caf_sliding_multiply_normalised(_p(d_x), d_x.size, _p(d_y), d_y.size, int(startIdx),
                                                           int(idxlen), coef, _p(d_pdts), None))
    return d_pdts


def multiTemplateSlidingDotProduct(d_x, d_templates, startIdx, idxlen, d_templateEnergies=None, numSlidesPerBlk=None,
                                   THREADS_PER_BLOCK=128):
    """ref: cupyExtensions.py:563-640, multiplySlices.cu:251-399.  Returns (templateIdx int32, qf2 float32)."""
    requireDtype(np.complex64, d_x)
    requireDtype(np.complex64, d_templates)
    if d_templates.ndim != 2:
        raise ValueError("Templates should be 2D; each row is an individual template.")
    numTemplates, templateLength = d_templates.shape
    if startIdx < 0:
        raise ValueError("startIdx should be >= 0.")
    endIdx = startIdx + idxlen - 1
    if endIdx + templateLength - 1 >= d_x.size:
        raise ValueError("final slide index (%d) should be within the bounds of d_x (%d)."
                         % (endIdx + templateLength - 1, d_x.size))
    if d_templateEnergies is None:
        t = d_templates.get().astype(np.complex128)
        d_templateEnergies = asarray(np.sum(np.abs(t) ** 2, axis=1).astype(np.float32))
    requireDtype(np.float32, d_templateEnergies)
    if d_templateEnergies.size != numTemplates:
        raise ValueError("d_templateEnergies size should be equal to the rows of templates.")
    if templateLength > 8192:
        raise MemoryError("x is too large to use this kernel.")
    d_templateIdx = empty(idxlen, np.int32)
    d_qf2 = empty(idxlen, np.float32)
    _lib.check(_lib.load().caf_multi_template_sliding_dot(_p(d_templates), _p(d_templateEnergies), numTemplates,
                                                          templateLength, _p(d_x), d_x.size, int(startIdx), int(idxlen),
                                                          _p(d_templateIdx), _p(d_qf2), None))
    return d_templateIdx, d_qf2


# ---- peak finding -------------------------------------------------------------------------
def cupyFindLocalMaxima(x, minHeight, numOutputPerBlk=32, THREADS_PER_BLK=32, maxNumPeaks=10000):
    """ref: cupyExtensions.py:651-686, peakfinding.cu:14-58.  Unlike the CUDA kernel (atomic
    compaction, nondeterministic order) the indices come back in ascending order."""
    requireDtype(np.float32, x)
    numPeaksFound = zeros(1, np.int32)
    peakIndex = zeros(maxNumPeaks, np.int32)
    _lib.check(_lib.load().caf_find_local_maxima(_p(x), x.size, float(minHeight), int(maxNumPeaks), _p(peakIndex),
                                                 _p(numPeaksFound), None))
    return peakIndex, numPeaksFound


def fftRows(d_x, inverse=False, out=None):
    """cp.fft.fft(d_x, axis=1) / cp.fft.ifft replacement (batched rocFFT rows, complex64)."""
    requireDtype(np.complex64, d_x)
    rows = d_x.shape[0] if d_x.ndim == 2 else 1
    length = d_x.shape[-1]
    if out is None:
        out = empty(d_x.shape, np.complex64)
    _lib.check(_lib.load().caf_fft_rows(_p(d_x), _p(out), rows, length, 1 if inverse else 0, None))
    _lib.check(_lib.load().caf_stream_sync(None))
    return out
