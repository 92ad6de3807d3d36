"""Chirp-Z transform objects with the reference's names (spectralRoutines.py:20-44, 239-391;
pybinds/ippCZT/CZT.cpp).  The cached Bluestein constants are computed once on the host in
float64 and cast to complex64 exactly as the reference does (spectralRoutines.py:335-351,
CZT.cpp:89-140); every transform runs on the GPU through ``caf_czt_run_many``
(x*aa -> rocFFT -> *fv -> inverse rocFFT -> slice*ww).
"""

import ctypes as ct

import numpy as np

from . import _lib
from .devarray import DeviceArray, asarray, empty, requireDeviceArray, requireDtype


def _smooth(n, max_prime):
    for p in (2, 3, 5, 7, 11, 13):
        if p > max_prime:
            break
        while n % p == 0:
            n //= p
    return n == 1


def next_fast_len(length, maxPrime=7):
    """ref: spectralRoutines.py:20-44, CZT.cpp:3-31."""
    n = int(length)
    while not _smooth(n, maxPrime):
        n += 1
    return n


def prev_fast_len(length, maxPrime=7):
    """ref: spectralRoutines.py:48-73."""
    n = int(length)
    while not _smooth(n, maxPrime):
        n -= 1
    return n


class _CZTBase:
    """Shared constants + GPU execution.  ``nfft_rule`` / ``w_rule`` reproduce the three upstream
    variants."""

    def __init__(self, xlength, f1, f2, binWidth, fs, nfft_rule, w_rule):
        self.binWidth = binWidth
        self.f1 = f1
        self.k = int((f2 - f1) / binWidth + 1)
        self.m = int(xlength)
        m, k = self.m, self.k
        self.nfft = next_fast_len(m + k + 1) if nfft_rule == "gpu" else next_fast_len(m + k - 1)
        wexp = binWidth / fs if w_rule == "cpp" else (f2 - f1 + binWidth) / (k * fs)
        kk = np.arange(-m + 1, max(k - 1, m - 1) + 1, dtype=np.float64)
        ww = np.exp(-2j * np.pi * wexp * (kk * kk / 2.0))
        fv = np.fft.fft(1.0 / ww[: k - 1 + m], self.nfft)
        nn = np.arange(m)
        aa = np.exp(2j * np.pi * f1 / fs * -nn.astype(np.float64)) * ww[m + nn - 1]
        self._ww64, self._fv64, self._aa64 = ww, fv, aa
        self.d_ww = asarray(ww.astype(np.complex64))
        self.d_fv = asarray(fv.astype(np.complex64))
        self.d_aa = asarray(aa.astype(np.complex64))
        self._d_wws = asarray(ww[m - 1 : m + k - 1].astype(np.complex64))

    def getFreq(self):
        return np.arange(self.k) * self.binWidth + self.f1

    def _run_dev(self, d_x, rows, out=None):
        if out is None:
            out = empty((rows, self.k), np.complex64)
        _lib.check(
            _lib.load().caf_czt_run_many(ct.c_void_p(d_x.ptr), rows, self.m, self.k, self.nfft, ct.c_void_p(self.d_aa.ptr),
                                         ct.c_void_p(self.d_fv.ptr), ct.c_void_p(self._d_wws.ptr), ct.c_void_p(out.ptr),
                                         None),
            "caf_czt_run_many",
        )
        return out


class CZTCachedGPU(_CZTBase):
    """ref: spectralRoutines.py:317-391.  Device arrays in, device arrays out."""

    def __init__(self, xlength, f1, f2, binWidth, fs):
        super().__init__(xlength, f1, f2, binWidth, fs, "gpu", "py")

    def run(self, x):
        requireDeviceArray(x)
        requireDtype(np.complex64, x)
        if x.ndim != 1 or x.size != self.m:
            raise ValueError("x must be 1D of length %d" % self.m)
        return self._run_dev(x, 1).reshape(self.k)

    def runMany(self, xmany, out=None):
        requireDeviceArray(xmany)
        requireDtype(np.complex64, xmany)
        if xmany.ndim != 2 or xmany.shape[1] != self.m:
            raise ValueError("xmany must be 2D with rows of length %d" % self.m)
        if out is not None:
            requireDtype(np.complex64, out)
            if out.shape != (xmany.shape[0], self.k):
                raise ValueError("out must have shape (%d, %d)" % (xmany.shape[0], self.k))
        res = self._run_dev(xmany, xmany.shape[0], out)
        return None if out is not None else res


class CZTCached(_CZTBase):
    """Host-array signature of the reference's CPU class (spectralRoutines.py:239-311); the
    transform itself runs on the GPU in complex64 (== ``convertTo32fc=True`` upstream)."""

    def __init__(self, xlength, f1, f2, binWidth, fs, convertTo32fc=False):
        super().__init__(xlength, f1, f2, binWidth, fs, "py", "py")
        cast = (lambda a: a.astype(np.complex64)) if convertTo32fc else (lambda a: a)
        self.ww, self.fv, self.aa = cast(self._ww64), cast(self._fv64), cast(self._aa64)

    def run(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        if x.ndim != 1 or x.size != self.m:
            raise ValueError("x must be 1D of length %d" % self.m)
        return self._run_dev(asarray(x), 1).get().reshape(self.k)

    def runMany(self, xmany, out=None):
        xmany = np.ascontiguousarray(xmany, dtype=np.complex64)
        res = self._run_dev(asarray(xmany), xmany.shape[0]).get()
        if out is None:
            return res
        out[...] = res


class pbIppCZT32fc(_CZTBase):
    """pybind class of pybinds/ippCZT (pbCZT.cpp:7-23, CZT.cpp:41-209): W exponent = fstep/fs,
    nfft = next_fast_len(len + k - 1); complex64 host arrays in and out."""

    def __init__(self, length, f1, f2, fstep, fs):
        super().__init__(length, f1, f2, fstep, fs, "py", "cpp")

    def run(self, x):
        x = np.ascontiguousarray(x, dtype=np.complex64)
        if x.ndim != 1 or x.size != self.m:
            raise ValueError("input length must be %d" % self.m)
        return self._run_dev(asarray(x), 1).get().reshape(self.k)

    def runMany(self, xmany):
        xmany = np.ascontiguousarray(xmany, dtype=np.complex64)
        if xmany.ndim != 2 or xmany.shape[1] != self.m:
            raise ValueError("input rows must have length %d" % self.m)
        return self._run_dev(asarray(xmany), xmany.shape[0]).get()


def czt(x, f1, f2, binWidth, fs):
    """One-shot CZT (ref: spectralRoutines.py:77-110) on the GPU; complex64 result."""
    x = np.asarray(x)
    return CZTCached(len(x), f1, f2, binWidth, fs, convertTo32fc=True).run(x)


def cupyDotTonesScaling(f0, fstep, numFreqs, src):
    """ref: spectralRoutines.py:580-630, genTones.cu:165-283 (`dotTonesScaling_32f`).  Dot products of the
    tones exp(j 2 pi (f0 + k fstep) i) with ``src`` in blocks of 64 samples; returns the
    (ceil(len / 64), numFreqs) complex64 interim array -- ``sum(axis=0)`` of it is the CZT of ``src`` at the
    normalised frequencies -(f0 + k fstep) (the upstream docstring's comparison)."""
    requireDeviceArray(src)
    requireDtype(np.complex64, src)
    length = src.size
    nblocks = (length + 63) // 64
    out = empty((nblocks, int(numFreqs)), np.complex64)
    _lib.check(_lib.load().caf_dot_tones(ct.c_void_p(src.ptr), length, float(f0), float(fstep), int(numFreqs),
                                         ct.c_void_p(out.ptr), None), "caf_dot_tones")
    return out


__all__ = ["next_fast_len", "prev_fast_len", "CZTCachedGPU", "CZTCached", "pbIppCZT32fc", "czt", "cupyDotTonesScaling",
           "DeviceArray"]
