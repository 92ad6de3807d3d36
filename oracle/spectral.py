"""Oracle (test infrastructure): frequency helpers and the Bluestein chirp-Z.

Restates, in NumPy, what the reference computes in
  * signalCreationRoutines.py:380-386   makeFreq
  * spectralRoutines.py:20-44           next_fast_len (7-smooth)
  * spectralRoutines.py:77-110          czt
  * spectralRoutines.py:239-311         CZTCached (.run / .runMany / .getFreq)
  * spectralRoutines.py:317-391         CZTCachedGPU (nfft rule differs, see gpu_rule)
  * pybinds/ippCZT/CZT.cpp:41-209       IppCZT32fc (cpp_rule: W exponent = fstep/fs)
Not product code.
"""

import numpy as np
import scipy.fft as _sfft


def makeFreq(length, fs):
    """FFT-order frequency vector; entries >= fs/2 wrap to negative.

    ref: signalCreationRoutines.py:380-386 (loop form).  Same arithmetic
    (i / length * fs, then subtract fs) so the values are bit-identical.
    """
    f = np.arange(length, dtype=np.float64) / length * fs
    wrap = f >= fs / 2
    f[wrap] = f[wrap] - fs
    return f


def _is_smooth(n, max_prime=7):
    for p in (2, 3, 5, 7, 11, 13):
        if p > max_prime:
            break
        while n % p == 0:
            n //= p
    return n == 1


def next_fast_len(length, maxPrime=7):
    """Smallest n >= length whose prime factors are all <= maxPrime.

    ref: spectralRoutines.py:20-44 (sympy.primefactors loop), CZT.cpp:3-31.
    """
    n = int(length)
    while not _is_smooth(n, maxPrime):
        n += 1
    return n


def dft(x, freqs, fs):
    """Brute-force DFT at arbitrary frequencies (ref: spectralRoutines.dft, used by
    tests/compare_czt_impl.py as the ground truth for the CZT)."""
    n = np.arange(len(x))
    return np.exp(-2j * np.pi * np.outer(np.asarray(freqs, dtype=np.float64), n) / fs) @ x


class CZTCached:
    """Cached-constants Bluestein CZT.

    Evaluates X(f_i) = sum_n x[n] exp(-j 2 pi f_i n / fs) on f_i = f1 + i*binWidth,
    i < k = int((f2-f1)/binWidth + 1), as  x*aa -> FFT_nfft -> *fv -> IFFT -> slice*ww.

    ref: spectralRoutines.py:239-311.  ``rule`` selects how nfft / the chirp
    rate are chosen, because the reference's three implementations differ
    (SURVEY Appendix B.8):
      "py"  : nfft = first 7-smooth >= m+k-1, W = (f2-f1+binWidth)/(k*fs)   (CZTCached)
      "gpu" : nfft = first 7-smooth >  m+k,   same W                         (CZTCachedGPU :326-333)
      "cpp" : nfft = next_fast_len(m+k-1),    W = binWidth/fs                (CZT.cpp:44,99)
    """

    def __init__(self, xlength, f1, f2, binWidth, fs, convertTo32fc=False, rule="py"):
        self.binWidth = binWidth
        self.f1 = f1
        self.k = int((f2 - f1) / binWidth + 1)
        self.m = int(xlength)
        if rule == "gpu":
            self.nfft = next_fast_len(self.m + self.k + 1)
        else:
            self.nfft = next_fast_len(self.m + self.k - 1)
        if rule == "cpp":
            wexp = binWidth / fs
        else:
            wexp = (f2 - f1 + binWidth) / (self.k * fs)
        m, k = self.m, self.k
        kk = np.arange(-m + 1, max(k - 1, m - 1) + 1, dtype=np.float64)
        self.ww = np.exp(-2j * np.pi * wexp * (kk * kk / 2.0))
        chirp = 1.0 / self.ww[: k - 1 + m]
        self.fv = np.fft.fft(chirp, self.nfft)
        nn = np.arange(m, dtype=np.float64)
        self.aa = np.exp(2j * np.pi * f1 / fs * -nn) * self.ww[m + np.arange(m) - 1]
        if convertTo32fc:
            self.ww = self.ww.astype(np.complex64)
            self.fv = self.fv.astype(np.complex64)
            self.aa = self.aa.astype(np.complex64)

    def getFreq(self):
        return np.arange(self.k) * self.binWidth + self.f1

    def run(self, x):
        return self.runMany(np.asarray(x)[None, :])[0]

    def runMany(self, xmany, out=None):
        m, k = self.m, self.k
        y = xmany * self.aa
        fy = _sfft.fft(y, self.nfft, axis=-1)  # scipy.fft keeps complex64
        fy = fy * self.fv
        g = _sfft.ifft(fy, axis=-1)
        res = g[..., m - 1 : m + k - 1] * self.ww[m - 1 : m + k - 1]
        if out is None:
            return res
        out[...] = res
        return out


def czt(x, f1, f2, binWidth, fs):
    """One-shot CZT (ref: spectralRoutines.py:77-110). nfft = 7-smooth >= m+k."""
    x = np.asarray(x)
    k = int((f2 - f1) / binWidth + 1)
    m = len(x)
    nfft = next_fast_len(m + k)
    kk = np.arange(-m + 1, max(k - 1, m - 1) + 1, dtype=np.float64)
    ww = np.exp(-2j * np.pi * (f2 - f1 + binWidth) / (k * fs) * (kk * kk / 2.0))
    fv = np.fft.fft(1.0 / ww[: k - 1 + m], nfft)
    nn = np.arange(m)
    aa = np.exp(2j * np.pi * f1 / fs * -nn.astype(np.float64)) * ww[m + nn - 1]
    g = np.fft.ifft(np.fft.fft(x * aa, nfft) * fv)
    return g[m - 1 : m + k - 1] * ww[m - 1 : m + k - 1]
