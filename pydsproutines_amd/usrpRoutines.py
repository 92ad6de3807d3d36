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
