"""ctypes binding of oracle/c/ippxcorrfft_port.c -- the plain-C, pthreads restatement of the
reference's threaded native correlator (IppXcorrFFT.cpp:13-52, 94-196; CyIppXcorrFFT.pyx:25-80).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): tests and bench.py's threaded CPU baseline.
"""

import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libippxcorrfft_port.so")
_lib = None


def build():
    """gcc build of the restatement (called by __graft_entry__.build())."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c")])
    return _SO


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ct.CDLL(_SO)
        f32 = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
        i32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        lib.ippxcorrfft_port.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_void_p, ct.c_int, ct.c_int, ct.c_int,
                                         ct.c_int, ct.c_int, f32, i32, ct.c_int]
        lib.ippxcorrfft_port.restype = ct.c_int
        _lib = lib
    return _lib


class IppXcorrFFT:
    """Same surface as oracle.xcorr.IppXcorrFFT (CyIppXcorrFFT.pyx:25-80): ctor (cutout c64, num_threads, autoConj);
    xcorr(rx c64, startIdx, endIdx, idxStep) -> (float32 QF^2, int32 bin)."""

    def __init__(self, cutout, num_threads=1, autoConj=True):
        if np.asarray(cutout).dtype != np.complex64:
            raise ValueError("cutout must be complex64")
        self.cutout = np.ascontiguousarray(cutout, dtype=np.complex64)
        self.num_threads = int(num_threads)
        self.autoConj = bool(autoConj)

    def xcorr(self, rx, startIdx, endIdx, idxStep):
        rx = np.ascontiguousarray(rx, dtype=np.complex64)
        n = len(range(startIdx, endIdx, idxStep))
        pk = np.zeros(n, np.float32)
        fi = np.zeros(n, np.int32)
        rc = load().ippxcorrfft_port(self.cutout.ctypes.data, self.cutout.size, int(self.autoConj), rx.ctypes.data,
                                     rx.size, int(startIdx), int(endIdx), int(idxStep), self.num_threads, pk, fi, n)
        if rc != 0:
            raise RuntimeError("ippxcorrfft_port returned %d" % rc)
        return pk, fi
