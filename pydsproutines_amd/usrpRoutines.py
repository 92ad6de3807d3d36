"""IQ ingest with the reference's reader signature (usrpRoutines.py:51-67 simpleBinRead): interleaved
int16 I/Q on disk -> complex64.  ``simpleBinRead`` keeps the reference's host-array contract;
``simpleBinReadToDevice`` copies the raw int16 (half the bytes of complex64) and converts on the GPU, the
pattern benchmarks/benchmark_cupyCopyAndConvert.py:17-25 measures."""

import ctypes as ct

import numpy as np

from . import _lib
from .devarray import asarray, empty


def iq16_to_complex64(d_iq16, scale=1.0):
    """Device int16 interleaved (2*n,) -> device complex64 (n,) * scale."""
    if d_iq16.dtype != np.int16:
        raise TypeError("Must be int16, found %s" % d_iq16.dtype)
    if d_iq16.size % 2:
        raise ValueError("interleaved IQ needs an even number of int16 values")
    n = d_iq16.size // 2
    out = empty(n, np.complex64)
    _lib.check(_lib.load().caf_iq16_to_c64(ct.c_void_p(d_iq16.ptr), n, float(scale), ct.c_void_p(out.ptr), None),
               "caf_iq16_to_c64")
    return out


class Iq16FrontEnd:
    """Front-end filter + decimate fused into the rx load: raw interleaved int16 IQ on the device ->
    ``lfilter(taps, 1, scale * iq)[dsPhase::dsr]`` as complex64 in one kernel, with the streaming state of the
    reference's ``CupyKernelFilter.run_filter_smtaps`` (filterRoutines.py:482-501): the tail of every chunk is
    carried into the next call, so chunked calls equal one long filter.  Equivalent to
    ``iq16_to_complex64`` followed by ``CupyKernelFilter.filter_smtaps(dsr=, dsPhase=)``.

    The decimation phase is tracked across chunks, so chunk lengths need not be multiples of ``dsr``."""

    def __init__(self, d_taps, dsr=1, dsPhase=0, scale=1.0):
        if d_taps.dtype != np.float32:
            raise TypeError("Must be float32, found %s" % d_taps.dtype)
        if d_taps.ndim != 1:
            raise ValueError("d_taps must be 1D.")
        if dsPhase >= dsr or dsPhase < 0:
            raise ValueError("dsPhase must be between in the range [0,dsr-1].")
        self.d_taps, self.dsr, self.scale = d_taps, int(dsr), float(scale)
        self.phase = int(dsPhase)  # phase of the next chunk
        self.delay = None  # int16 (2 * (ntaps - 1),): the IQ pairs preceding the next chunk

    def reset(self, dsPhase=0):
        self.phase, self.delay = int(dsPhase), None

    def run(self, d_iq16):
        if d_iq16.dtype != np.int16:
            raise TypeError("Must be int16, found %s" % d_iq16.dtype)
        if d_iq16.ndim != 1 or d_iq16.size % 2 or d_iq16.size == 0:
            raise ValueError("interleaved IQ needs a 1D, non-empty, even number of int16 values")
        n = d_iq16.size // 2
        nout = max(0, (n - self.phase + self.dsr - 1) // self.dsr)
        out = empty(nout, np.complex64)
        lib = _lib.load()
        dl = self.delay.size // 2 if self.delay is not None else 0
        _lib.check(lib.caf_iq16_fir_decimate(ct.c_void_p(d_iq16.ptr), n, self.scale, ct.c_void_p(self.d_taps.ptr),
                                             self.d_taps.size, ct.c_void_p(self.delay.ptr) if dl else None, dl, self.dsr,
                                             self.phase, ct.c_void_p(out.ptr), nout, None), "caf_iq16_fir_decimate")
        # carry the last ntaps-1 input pairs (old history + this chunk) and the phase into the next call
        keep = self.d_taps.size - 1
        if keep > 0:
            nd = empty(2 * keep, np.int16)
            from_x = min(keep, n)
            from_old = keep - from_x
            if from_old:
                if dl >= from_old:
                    _lib.check(lib.caf_d2d(ct.c_void_p(nd.ptr), ct.c_void_p(self.delay.ptr + 4 * (dl - from_old)),
                                           4 * from_old, None))
                else:  # not enough history yet: zeros in front
                    _lib.check(lib.caf_memset(ct.c_void_p(nd.ptr), 0, 4 * (from_old - dl), None))
                    if dl:
                        _lib.check(lib.caf_d2d(ct.c_void_p(nd.ptr + 4 * (from_old - dl)), ct.c_void_p(self.delay.ptr),
                                               4 * dl, None))
            _lib.check(lib.caf_d2d(ct.c_void_p(nd.ptr + 4 * from_old), ct.c_void_p(d_iq16.ptr + 4 * (n - from_x)),
                                   4 * from_x, None))
            self.delay = nd
        # next kept sample index relative to the next chunk's start
        self.phase = self.phase + nout * self.dsr - n
        return out


def simpleBinReadToDevice(filename, numSamps=-1, in_dtype=np.int16, offset=0, scale=1.0):
    """Read numSamps complex samples of interleaved int16 and return them as a complex64 DeviceArray."""
    if in_dtype != np.int16:
        raise TypeError("device-side conversion is implemented for int16 recordings")
    raw = np.fromfile(filename, dtype=np.int16, count=numSamps * 2 if numSamps >= 0 else -1, offset=offset)
    return iq16_to_complex64(asarray(raw), scale)


def simpleBinRead(filename, numSamps=-1, in_dtype=np.int16, out_dtype=np.complex64, offset=0):
    """Reference signature and return type (host complex array); int16 -> complex64 is converted on the GPU."""
    if in_dtype == np.complex64 or in_dtype == np.complex128:
        raise TypeError("in_dtype must be a real type. You likely want float32 or float64 instead.")
    if in_dtype == np.int16 and out_dtype == np.complex64:
        return simpleBinReadToDevice(filename, numSamps, in_dtype, offset).get()
    # other on-disk types are plain host reinterpretation (no arithmetic involved)
    data = np.fromfile(filename, dtype=in_dtype, count=numSamps * 2 if numSamps >= 0 else -1, offset=offset)
    return data.astype(np.float32 if out_dtype == np.complex64 else np.float64).view(out_dtype)


def multiBinRead(filenames, numSamps, in_dtype=np.int16, out_dtype=np.complex64, offset=0):
    """ref: usrpRoutines.py:70-84.  Concatenated recordings, numSamps complex samples of each."""
    alldata = np.zeros(len(filenames) * numSamps, out_dtype)
    for i, fn in enumerate(filenames):
        alldata[i * numSamps : (i + 1) * numSamps] = simpleBinRead(fn, numSamps, in_dtype, out_dtype, offset=offset)
    return alldata


def multiBinReadThreaded(filenames, numSamps, in_dtype=np.int16, out_dtype=np.complex64, offset=0, threads=2):
    """ref: usrpRoutines.py:87-114.  The disk reads run in ``threads`` workers; the int16 -> complex64
    conversions are issued from the calling thread (one device context) as the raw buffers arrive."""
    import concurrent.futures

    if in_dtype == np.complex64 or in_dtype == np.complex128:
        raise TypeError("in_dtype must be a real type. You likely want float32 or float64 instead.")
    alldata = np.zeros(len(filenames) * numSamps, out_dtype)
    with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as executor:
        futs = {futureBinRead(executor, fn, numSamps, in_dtype, offset): i for i, fn in enumerate(filenames)}
        for fut in concurrent.futures.as_completed(futs):
            i = futs[fut]
            raw = fut.result()
            if in_dtype == np.int16 and out_dtype == np.complex64:
                alldata[i * numSamps : (i + 1) * numSamps] = iq16_to_complex64(asarray(raw)).get()
            else:
                alldata[i * numSamps : (i + 1) * numSamps] = raw.astype(
                    np.float32 if out_dtype == np.complex64 else np.float64).view(out_dtype)
    return alldata


def futureBinRead(executor, filename, numSamps, in_dtype=np.int16, offset=0):
    """ref: usrpRoutines.py:117-156.  Submits the raw read (2 * numSamps values of ``in_dtype``) to an existing
    ThreadPoolExecutor; ``future.result()`` is the interleaved array, ready for ``iq16_to_complex64(asarray(.))``
    while the next file is already loading."""
    if in_dtype == np.complex64 or in_dtype == np.complex128:
        raise TypeError("in_dtype must be a real type. You likely want float32 or float64 instead.")
    return executor.submit(np.fromfile, filename, dtype=in_dtype, count=numSamps * 2, offset=offset)
