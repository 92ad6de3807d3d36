"""Synthetic-signal helpers the xcorr benchmarks use (host side, NumPy): same names, arguments and
return values as the reference's signalCreationRoutines.py (makeFreq :380-386, randBits :20-21,
symsFromBits :24-43, randPSKsyms :47-69, randnoise :72-104, addSigToNoise :107-145)."""

import numpy as np


def makeFreq(length, fs):
    """FFT-order frequency vector with entries >= fs/2 wrapped to negative frequencies."""
    f = np.arange(int(length), dtype=np.float64) / length * fs
    hi = f >= fs / 2
    f[hi] -= fs
    return f


def randBits(length, m):
    return np.random.randint(0, m, length, dtype=np.uint8)


def symsFromBits(bits, m, dtype=np.complex128):
    if m not in (2, 4, 8):
        raise KeyError(m)
    table = np.exp(2j * np.pi * np.arange(m) / m)
    table = np.round(table * 1e15) / 1e15 if m in (2, 4) else table  # exact +-1, +-1j for BPSK/QPSK
    return table.astype(dtype)[bits]


def randPSKsyms(length, m, dtype=np.complex128):
    """(symbols, bits) of random m-ary PSK."""
    bits = randBits(length, m)
    return symsFromBits(bits, m, dtype), bits


def randnoise(length, bw_signal, chnBW, snr_inband_linear, sigPwr=1.0):
    """Complex Gaussian noise scaled for a target in-band SNR."""
    base = (np.random.randn(length) + 1j * np.random.randn(length)) / np.sqrt(2) * np.sqrt(sigPwr)
    return base * np.sqrt(1.0 / snr_inband_linear) * np.sqrt(chnBW / bw_signal)


def addSigToNoise(signal, noiseLen=None, sigStartIdx=0, bw_signal=1, chnBW=1, snr_inband_linear=np.inf, sigPwr=1.0,
                  fshift=None):
    """Embed `signal` in noise at `sigStartIdx`, optionally frequency-shifting the sum.
    Returns (noise, rx) or (noise, rx, tone)."""
    if noiseLen is None:
        noiseLen = len(signal)
    if snr_inband_linear is np.inf:
        noise = np.zeros(noiseLen, dtype=np.complex128)
    else:
        noise = randnoise(noiseLen, bw_signal, chnBW, snr_inband_linear, sigPwr)
    rx = np.zeros(noiseLen, dtype=np.complex128)
    rx[sigStartIdx : len(signal) + sigStartIdx] = signal
    rx = rx + noise
    if fshift is not None:
        tone = np.exp(1j * 2 * np.pi * fshift * np.arange(noiseLen) / chnBW)
        return noise, rx * tone, tone
    return noise, rx


def addManySigToNoise(noiseLen, sigStartIdxList, signalList, bw_signal, chnBW, snr_inband_linearList, fshifts=None,
                      sigStartTimeList=None):
    """ref: signalCreationRoutines.py:148-218 (what benchmark_multiTemplateDotKernels.py:42-48 builds its input with): one
    noise record scaled for the FIRST signal's SNR, every unit-power signal placed at its start index with amplitude
    sqrt(snr_i / snr_0), optionally frequency-shifted.  Returns (noise, rx) or (noise, rx, tones).  The sub-sample placement
    (sigStartTimeList, upstream's propagateSignal) is outside the CAF path and not provided."""
    if sigStartTimeList is not None:
        raise NotImplementedError("sub-sample placement (sigStartTimeList) is outside the CAF hot path")
    snrs = np.asarray(snr_inband_linearList, dtype=np.float64)
    noise = randnoise(noiseLen, bw_signal, chnBW, snrs[0], 1.0)
    parts = np.zeros((len(snrs), noiseLen), dtype=np.complex128)
    for i, (start, sig) in enumerate(zip(sigStartIdxList, signalList)):
        parts[i, start : start + len(sig)] = np.asarray(sig) * np.sqrt(snrs[i] / snrs[0])
    if fshifts is None:
        return noise, parts.sum(axis=0) + noise
    tones = np.exp(2j * np.pi * np.asarray(fshifts, dtype=np.float64)[:, None] * np.arange(noiseLen) / chnBW)
    return noise, (parts * tones).sum(axis=0) + noise, tones
