"""Oracle (test infrastructure): cross-correlation / CAF flavours.

NumPy restatement of the reference's CPU algorithms.  The reference evaluates
everything in a per-delay Python loop; here each flavour is written as a
chunked, batched array expression over a (delays x window) view so that the
oracle finishes in seconds at test sizes.  The arithmetic per delay is the
reference's:  product with the conjugated template -> DFT over frequency ->
|.|^2 -> divide by template energy and by the sliding rx energy.

Reference locations (relative to the upstream repository root):
  fastXcorr                xcorrRoutines.py:460-580   (six branches A/A'/B/B'/C/C')
  cztXcorr                 xcorrRoutines.py:413-457
  GroupXcorr               xcorrRoutines.py:852-954
  GroupXcorrCZT            xcorrRoutines.py:957-1039
  GroupXcorrFFT            xcorrRoutines.py:1047-1189 (CPU .xcorr)
  TemplateCrossCorrelator  xcorrRoutines.py:277-371   (cupy calls -> numpy)
  cp_fastXcorr / _v2       xcorrRoutines.py:29-274    (output dtypes / order)
  IppXcorrFFT_32fc         cython_ext/CyIppXcorrFFT/IppXcorrFFT.cpp:94-194
  GroupXcorrFFT (C++)      cython_ext/CyGroupXcorrFFT/GroupXcorrFFT.cpp:3-203
Not product code.  Energies are accumulated in float64 (the reference uses
float32 NumPy norms on the Python path and float64 IPP norms on the native
path); the difference is ~1e-7 relative and is covered by the pin tolerance.
"""

import numpy as np
import scipy.fft as _sfft
from numpy.lib.stride_tricks import sliding_window_view

from .spectral import CZTCached, makeFreq

_CHUNK_ELEMS = 1 << 22  # elements of the (delays x N) working matrix per chunk


def _energy(x, axis=-1):
    x = np.asarray(x)
    return np.sum(x.real.astype(np.float64) ** 2 + x.imag.astype(np.float64) ** 2, axis=axis)


def _windows(rx, n, shifts):
    """Rows rx[s:s+n] for each s in shifts (a copy, shape (len(shifts), n))."""
    view = sliding_window_view(rx, n)
    return view[np.asarray(shifts, dtype=np.int64)]


def _chunks(num_rows, row_len):
    step = max(1, _CHUNK_ELEMS // max(1, row_len))
    for a in range(0, num_rows, step):
        yield a, min(num_rows, a + step)


def argmax2d(m):
    """ref: xcorrRoutines.py:815-830."""
    return np.unravel_index(np.argmax(m), m.shape)


def calcQF2(x, y):
    """ref: xcorrRoutines.py:833-848."""
    x = np.asarray(x)
    y = np.asarray(y)
    if x.ndim == 1:
        return np.abs(np.vdot(x, y)) ** 2 / _energy(x) / _energy(y)
    return np.abs(np.sum(x * y.conj(), axis=1)) ** 2 / _energy(x, 1) / _energy(y, 1)


def fastXcorr(cutout, rx, freqsearch=False, outputCAF=False, shifts=None, absResult=True):
    """All six branches of the reference's fastXcorr (xcorrRoutines.py:460-580).

    Output dtypes follow the reference: float64 / complex128 values, uint32
    frequency-bin indices.  Branch A' is sum(conj(rx)*cutout) (np.vdot order,
    :503) i.e. the conjugate of the other complex branches (Appendix B.5).
    """
    cutout = np.asarray(cutout)
    rx = np.asarray(rx)
    n = len(cutout)
    if shifts is None:
        shifts = np.arange(len(rx) - n + 1)
    shifts = np.asarray(shifts)
    ns = len(shifts)
    e_c = _energy(cutout)
    cconj = cutout.conj()

    if not freqsearch:
        out = np.zeros(ns, dtype=np.float64 if absResult else np.complex128)
    elif not outputCAF:
        out = np.zeros(ns, dtype=np.float64 if absResult else np.complex128)
        fidx = np.zeros(ns, dtype=np.uint32)
    else:
        out = np.zeros((ns, n), dtype=np.float64 if absResult else np.complex128)

    for a, b in _chunks(ns, n):
        w = _windows(rx, n, shifts[a:b])
        e_w = _energy(w, 1)
        if not freqsearch:
            d = (w.conj() * cutout).sum(axis=1, dtype=np.complex128)
            if absResult:
                out[a:b] = np.abs(d) ** 2 / e_c / e_w
            else:
                out[a:b] = d / np.sqrt(e_c) / np.sqrt(e_w)
            continue
        spec = _sfft.fft(w * cconj, axis=1)  # stays complex64 for complex64 input
        if outputCAF:
            if absResult:
                out[a:b] = (np.abs(spec).astype(np.float64) ** 2) / e_c / e_w[:, None]
            else:
                out[a:b] = spec / np.sqrt(e_c) / np.sqrt(e_w)[:, None]
        else:
            mag = np.abs(spec)
            im = np.argmax(mag, axis=1)
            fidx[a:b] = im
            pk = spec[np.arange(b - a), im]
            if absResult:
                out[a:b] = np.abs(pk).astype(np.float64) ** 2 / e_c / e_w
            else:
                out[a:b] = pk / np.sqrt(e_c) / np.sqrt(e_w)
    if freqsearch and not outputCAF:
        return out, fidx
    return out


def caf_bins(cutout, rx, bins, shifts=None):
    """CAF restricted to a subset of the N-bin FFT grid: the columns ``bins``
    (may be negative, taken mod N) of fastXcorr(..., freqsearch=True,
    outputCAF=True).  This is the literal per-delay algorithm the hypothesis
    engine must reproduce.  Returns float64 (S, F).
    """
    cutout = np.asarray(cutout)
    rx = np.asarray(rx)
    n = len(cutout)
    if shifts is None:
        shifts = np.arange(len(rx) - n + 1)
    shifts = np.asarray(shifts)
    cols = np.mod(np.asarray(bins, dtype=np.int64), n)
    e_c = _energy(cutout)
    cconj = cutout.conj()
    out = np.zeros((len(shifts), len(cols)), dtype=np.float64)
    for a, b in _chunks(len(shifts), n):
        w = _windows(rx, n, shifts[a:b])
        spec = _sfft.fft(w * cconj, axis=1)[:, cols]
        out[a:b] = (np.abs(spec).astype(np.float64) ** 2) / e_c / _energy(w, 1)[:, None]
    return out


def caf_overlap_save(cutout, rx, bins, block=1 << 16, workers=1):
    """Same-algorithm CPU baseline (baseline 3): the hypothesis-domain
    overlap-save CAF the GPU runs, via scipy.fft.  For each rx block X=FFT(rx_b),
    hypothesis k: IFFT(X * conj(H0[(m - k*B/N) mod B])).  Returns float32 (S, F).
    Only valid for N | B (power-of-two N).
    """
    cutout = np.asarray(cutout, dtype=np.complex64)
    rx = np.asarray(rx, dtype=np.complex64)
    n = len(cutout)
    bsz = int(block)
    assert bsz % n == 0 and bsz > n
    s_total = len(rx) - n + 1
    step = bsz - n + 1
    h0c = np.conj(_sfft.fft(cutout, bsz)).astype(np.complex64)
    e_c = _energy(cutout)
    csum = np.concatenate(([0.0], np.cumsum(_energy(rx[:, None], 1))))
    e_w = csum[n:] - csum[:-n]
    bins = np.asarray(bins, dtype=np.int64)
    out = np.empty((s_total, len(bins)), dtype=np.float32)
    for s0 in range(0, s_total, step):
        seg = rx[s0 : s0 + bsz]
        if len(seg) < bsz:
            seg = np.concatenate((seg, np.zeros(bsz - len(seg), np.complex64)))
        xf = _sfft.fft(seg, workers=workers)
        nv = min(step, s_total - s0)
        hyp = np.stack([xf * np.roll(h0c, int(k) * (bsz // n)) for k in bins])
        r = _sfft.ifft(hyp, axis=1, workers=workers)[:, :nv]
        out[s0 : s0 + nv] = ((r.real**2 + r.imag**2).T / (e_c * e_w[s0 : s0 + nv, None])).astype(np.float32)
    return out


def cztXcorr(cutout, rx, f_searchMin, f_searchMax, fs, cztStep=0.1, outputCAF=False, shifts=None):
    """ref: xcorrRoutines.py:413-457.  outputCAF -> (float64 (S,k), freqs);
    else (rx.dtype (S,) complex QF at the peak bin, float64 (S,) peak freq in Hz)."""
    cutout = np.asarray(cutout)
    rx = np.asarray(rx)
    n = cutout.size
    cz = CZTCached(n, f_searchMin, f_searchMax, cztStep, fs)
    freqs = cz.getFreq()
    if shifts is None:
        shifts = np.arange(len(rx) - n + 1)
    shifts = np.asarray(shifts)
    e_c = _energy(cutout)
    cconj = cutout.conj()
    if outputCAF:
        res = np.zeros((len(shifts), freqs.size))
    else:
        res = np.zeros(shifts.size, dtype=rx.dtype)
        fpk = np.zeros(shifts.size, dtype=np.float64)
    for a, b in _chunks(len(shifts), 4 * cz.nfft):
        w = _windows(rx, n, shifts[a:b])
        e_w = _energy(w, 1)
        z = cz.runMany(w * cconj)
        if outputCAF:
            res[a:b] = np.abs(z) ** 2.0 / e_w[:, None] / e_c
        else:
            mi = np.argmax(np.abs(z), axis=1)
            res[a:b] = z[np.arange(b - a), mi] / np.sqrt(e_w) / np.sqrt(e_c)
            fpk[a:b] = freqs[mi]
    if outputCAF:
        return res, freqs
    return res, fpk


class GroupXcorr:
    """Composite (grouped) template against an explicit frequency list by dense DFT.

    ref: xcorrRoutines.py:852-954.  Note the reference builds the DFT phases from
    the ORIGINAL (un-zeroed) ``starts`` (:907-915; Appendix B.6) -- a constant
    phase per frequency, invisible after abs -- and returns peak frequencies in
    Hz, not indices (:952).
    """

    def __init__(self, y, starts, lengths, freqs, fs, autoConj=True, autoZeroStarts=True):
        y = np.asarray(y)
        starts = np.asarray(starts)
        lengths = np.asarray(lengths)
        assert starts.size == lengths.size
        self.starts = starts - starts[0] if autoZeroStarts else starts
        self.lengths = lengths
        self.numGroups = starts.size
        self.freqs = np.asarray(freqs, dtype=np.float64)
        self.fs = fs
        idx = np.concatenate([np.arange(s, s + l) for s, l in zip(starts, lengths)])
        self._rel = np.concatenate([np.arange(s, s + l) for s, l in zip(self.starts, lengths)])
        self.yconcat = y.conj()[idx] if autoConj else y[idx]
        self.yconcatNormSq = _energy(self.yconcat)
        self.freqMat = np.exp(-2j * np.pi * self.freqs[:, None] * idx[None, :] / fs)

    def xcorr(self, rx, shifts=None):
        rx = np.asarray(rx)
        span = int(self.starts[-1] + self.lengths[-1])
        if shifts is None:
            shifts = np.arange(len(rx) - span + 1)
        else:
            shifts = np.asarray(shifts)
            assert shifts[-1] + span < rx.size
        xc = np.zeros(shifts.size)
        fpk = np.zeros(shifts.size)
        ltot = self._rel.size
        for a, b in _chunks(shifts.size, max(ltot, self.freqs.size) * 4):
            g = rx[shifts[a:b, None] + self._rel[None, :]]
            e = _energy(g, 1)
            pf = (g * self.yconcat) @ self.freqMat.T  # complex128
            mag = np.abs(pf)
            im = np.argmax(mag, axis=1)
            xc[a:b] = mag[np.arange(b - a), im] ** 2 / e / self.yconcatNormSq
            fpk[a:b] = self.freqs[im]
        return xc, fpk

    def caf(self, rx, shifts):
        """Full (S, F) QF^2 plane (not a reference method; used to check surfaces)."""
        rx = np.asarray(rx)
        shifts = np.asarray(shifts)
        g = rx[shifts[:, None] + self._rel[None, :]]
        pf = (g * self.yconcat) @ self.freqMat.T
        return np.abs(pf) ** 2 / _energy(g, 1)[:, None] / self.yconcatNormSq


class GroupXcorrCZT:
    """ref: xcorrRoutines.py:957-1039.  Returns (float64 (S,k), cztFreq)."""

    def __init__(self, y, starts, lengths, f1, f2, binWidth, fs, autoConj=True, autoZeroStarts=True):
        y = np.asarray(y)
        starts = np.asarray(starts)
        lengths = np.asarray(lengths)
        assert starts.size == lengths.size
        self.starts = starts - starts[0] if autoZeroStarts else starts
        self.lengths = lengths
        self.numGroups = starts.size
        self.fs = fs
        self.f1, self.f2, self.binWidth = f1, f2, binWidth
        self.maxLength = int(np.max(lengths))
        self.ystack = np.zeros((self.numGroups, self.maxLength), y.dtype)
        for i in range(self.numGroups):
            self.ystack[i, : lengths[i]] = y[starts[i] : starts[i] + lengths[i]]
        if autoConj:
            self.ystack = self.ystack.conj()
        self.ystackNormSq = _energy(self.ystack.ravel())
        self.cztc = CZTCached(self.maxLength, f1, f2, binWidth, fs)

    def xcorr(self, rx, shifts=None):
        rx = np.asarray(rx)
        span = int(self.starts[-1] + self.lengths[-1])
        if shifts is None:
            shifts = np.arange(len(rx) - span + 1)
        else:
            shifts = np.asarray(shifts)
            assert shifts[-1] + span < rx.size
        nf = int((self.f2 - self.f1) / self.binWidth + 1)
        cztFreq = np.arange(self.f1, self.f2 + self.binWidth / 2, self.binWidth)
        phases = np.exp(-2j * np.pi * cztFreq * self.starts.reshape((-1, 1)) / self.fs)
        xc = np.zeros((shifts.size, nf))
        for i, s in enumerate(shifts):
            acc = np.zeros(cztFreq.size, np.complex128)
            e = 0.0
            for g in range(self.numGroups):
                lg = int(self.lengths[g])
                seg = rx[s + self.starts[g] : s + self.starts[g] + lg]
                e += _energy(seg)
                pdt = np.zeros(self.maxLength, dtype=seg.dtype)
                pdt[:lg] = self.ystack[g, :lg] * seg
                acc += self.cztc.run(pdt) * phases[g]
            xc[i] = np.abs(acc) ** 2 / e / self.ystackNormSq
        return xc, cztFreq


def makeTimeScanSteervec(td_scan_range, fs, siglen):
    """ref: xcorrRoutines.py:670-676."""
    return np.exp(1j * 2 * np.pi * makeFreq(siglen, fs) * np.asarray(td_scan_range).reshape((-1, 1)))


def _time_scan(x, y, steeringvec, sigFreq, bounds):
    """ref: xcorrRoutines.py:644-667 / 696-719 (shared by fineFreqTimeSearch and GenXcorr.xcorr)."""
    x_fft = np.fft.fft(x)
    y_fft = np.fft.fft(y)
    rx_vec = x_fft * y_fft.conj()
    if bounds is not None:
        rx_vec[np.logical_or(sigFreq < bounds[0], sigFreq >= bounds[1])] = 0
    return np.dot(rx_vec, steeringvec.conj().T) / np.linalg.norm(x_fft) / np.linalg.norm(y_fft)


def fineFreqTimeSearch(x_aligned, y_aligned, fineRes, freqfound, freqRes, fs, td_scan_range, steeringvec=None,
                       td_scan_freqBounds=None):
    """ref: xcorrRoutines.py:583-667.  Returns (finefreqfound | None, timediff, cost_vec)."""
    x_aligned = np.asarray(x_aligned)
    y_aligned = np.asarray(y_aligned)
    n = len(x_aligned)
    if len(fineRes) > 0:
        precomputed = y_aligned.conj() * x_aligned
        for i in range(len(fineRes)):
            fineFreq = np.arange(freqfound - freqRes, freqfound + freqRes, fineRes[i])
            fineshifts = np.exp(1j * 2 * np.pi * -fineFreq.reshape((-1, 1)) * np.arange(n) / fs)
            pp = np.array([np.vdot(precomputed, fineshifts[j]) for j in range(len(fineFreq))])
            ind = int(np.argmax(np.abs(pp)))
            freqfound = fineFreq[ind]
        finefreqfound = freqfound
        x_aligned = x_aligned * fineshifts[ind]
    else:
        finefreqfound = None
    td_scan_range = np.asarray(td_scan_range)
    if steeringvec is None:
        steeringvec = makeTimeScanSteervec(td_scan_range, fs, n)
    cost_vec = _time_scan(x_aligned, y_aligned, steeringvec, makeFreq(n, fs), td_scan_freqBounds)
    return finefreqfound, td_scan_range[int(np.argmax(np.abs(cost_vec)))], cost_vec


class GenXcorr:
    """ref: xcorrRoutines.py:679-719."""

    def __init__(self, td_scan_range, fs, siglen):
        self.td_scan_range = np.asarray(td_scan_range)
        self.fs = fs
        self.sigFreq = makeFreq(siglen, fs)
        self.steeringvec = np.exp(1j * 2 * np.pi * self.sigFreq * self.td_scan_range.reshape((-1, 1)))
        self.td_scan_freqBounds = None

    def setTDscan_freqBounds(self, td_scan_freqBounds):
        self.td_scan_freqBounds = td_scan_freqBounds

    def xcorr(self, x, y):
        cost_vec = _time_scan(np.asarray(x), np.asarray(y), self.steeringvec, self.sigFreq, self.td_scan_freqBounds)
        return self.td_scan_range[int(np.argmax(np.abs(cost_vec)))], cost_vec


class GroupXcorrGPU(GroupXcorr):
    """ref: xcorrRoutines.py:1897-2058 (defined under `import cupy` upstream: restated from the source
    text).  ``xcorr`` is the parent's; ``xcorrKernel`` returns (QF^2 float32[S], frequency index int32[S])."""

    def __init__(self, y, starts, lengths, freqs, fs):
        super().__init__(y, starts, lengths, freqs, fs)

    def xcorrKernel(self, rx, shifts, numShiftsPerBlk=2, verbTiming=False):
        shifts = np.asarray(shifts)
        assert shifts.size % numShiftsPerBlk == 0
        caf = self.caf(rx, shifts)
        idx = np.argmax(caf, axis=1)
        return caf[np.arange(shifts.size), idx].astype(np.float32), idx.astype(np.int32)


class GroupXcorrCZT_Permutations:
    """ref: xcorrRoutines.py:1264-1690 (defined under `import cupy` upstream: restated from the source
    text of the CPU methods ``xcorr`` :1486-1547 / ``_xcorrThread`` :1635-1690 / ``getCAF`` :1549-1633).
    Pinned by the identity getCAF(selection) == GroupXcorrCZT(composite of the selected templates).xcorr,
    checked against the importable reference class in tests/golden/make_golden.py."""

    def __init__(self, ygroups, ygroupIdxs, groupStarts, f1, f2, binWidth, fs, autoConj=True):
        ygroups = np.asarray(ygroups)
        ygroupIdxs = np.asarray(ygroupIdxs)
        groupStarts = np.asarray(groupStarts)
        assert ygroups.shape[0] == ygroupIdxs.size
        assert np.unique(ygroupIdxs).size == groupStarts.size
        self.numTemplates = ygroupIdxs.size
        self.numGroups = groupStarts.size
        assert np.all(np.sort(np.unique(ygroupIdxs)) == np.arange(self.numGroups))
        self.groupStarts, self.ygroupIdxs, self.fs = groupStarts, ygroupIdxs, fs
        self.length = ygroups.shape[1]
        self.f1, self.f2, self.binWidth = f1, f2, binWidth
        self.ygroups = ygroups.conj() if autoConj else ygroups
        self.ygroupsEnergy = np.linalg.norm(self.ygroups, axis=1) ** 2

    def xcorr(self, rx, shifts=None, numThreads=1):
        rx = np.asarray(rx)
        if shifts is None:
            shifts = np.arange(len(rx) - (self.groupStarts[-1] + self.length) + 1)
        else:
            shifts = np.asarray(shifts)
            assert shifts[-1] + self.groupStarts[-1] + self.length < rx.size
        k = int((self.f2 - self.f1) / self.binWidth + 1)
        self.xcTemplates = np.zeros((self.numTemplates, shifts.size, k), dtype=np.complex128)
        self.rxgroupNormSq = np.zeros((self.numGroups, shifts.size))
        cztFreq = np.arange(self.f1, self.f2 + self.binWidth / 2, self.binWidth)
        groupPhases = np.exp(-1j * 2 * np.pi * cztFreq * self.groupStarts.reshape((-1, 1)) / self.fs)
        cztc = CZTCached(self.length, self.f1, self.f2, self.binWidth, self.fs)
        for i, shift in enumerate(shifts):
            for t in range(self.numTemplates):
                g = self.ygroupIdxs[t]
                rxgroup = rx[shift + self.groupStarts[g] : shift + self.groupStarts[g] + self.length]
                self.rxgroupNormSq[g, i] = _energy(rxgroup)
                self.xcTemplates[t, i, :] = cztc.run(self.ygroups[t, :] * rxgroup) * groupPhases[g, :]
        return cztFreq

    def getCAF(self, templateIdx, numThreads=4):
        templateIdx = np.asarray(templateIdx)
        assert templateIdx.size == self.numGroups
        cafcplx = np.zeros(self.xcTemplates.shape[1:], dtype=np.complex128)
        rxnormsq = np.zeros(self.rxgroupNormSq.shape[1])
        ynormsq = 0.0
        for g in range(templateIdx.size):
            t = np.argwhere(self.ygroupIdxs == g)[templateIdx[g]][0]
            cafcplx += self.xcTemplates[t]
            rxnormsq += self.rxgroupNormSq[g, :]
            ynormsq += self.ygroupsEnergy[t]
        return np.abs(cafcplx) ** 2 / rxnormsq.reshape(-1, 1) / ynormsq


class GroupXcorrFFT:
    """Equal-length groups on the makeFreq(fftlen, fs) grid.

    ref: xcorrRoutines.py:1047-1189 (lives inside ``try: import cupy`` upstream,
    so it is restated from source text; pinned by the equivalence
    GroupXcorrFFT == GroupXcorr(freqs=makeFreq(fftlen, fs))).
    """

    def __init__(self, ygroups, starts, fs, autoConj=True, fftlen=None, autoZeroStarts=True):
        ygroups = np.asarray(ygroups)
        starts = np.asarray(starts)
        assert starts.size == ygroups.shape[0]
        self.starts = starts - starts[0] if autoZeroStarts else starts
        self.numGroups = starts.size
        self.fs = fs
        self.ygroupLen = ygroups.shape[1]
        self.fftlen = self.ygroupLen if fftlen is None else int(fftlen)
        self.ygroupNormSq = _energy(ygroups.ravel())
        self.ygroups = ygroups.conj() if autoConj else ygroups
        self.fftfreq = makeFreq(self.fftlen, fs)
        self.groupPhases = np.exp(-2j * np.pi * self.fftfreq * self.starts.reshape((-1, 1)) / fs)

    def xcorr(self, rx, shifts=None, flattenToTime=True):
        rx = np.asarray(rx)
        if shifts is None:
            shifts = np.arange(len(rx) - (self.starts[-1] + self.fftlen) + 1)
        else:
            shifts = np.asarray(shifts)
            assert shifts[-1] + self.starts[-1] + self.fftlen < rx.size
        L = self.ygroupLen
        rel = (self.starts[:, None] + np.arange(L)[None, :])  # (G, L)
        if flattenToTime:
            xc = np.zeros(shifts.size)
            fi = np.zeros(shifts.size, dtype=np.uint32)
        else:
            xc = np.zeros((shifts.size, self.fftlen))
        for a, b in _chunks(shifts.size, self.numGroups * self.fftlen):
            g = rx[shifts[a:b, None, None] + rel[None, :, :]]  # (s, G, L)
            e = _energy(g.reshape(b - a, -1), 1)
            spec = np.fft.fft(g * self.ygroups[None], n=self.fftlen, axis=2)
            comb = np.sum(spec * self.groupPhases[None], axis=1)
            ff = np.abs(comb) ** 2 / e[:, None] / self.ygroupNormSq
            if flattenToTime:
                im = np.argmax(ff, axis=1)
                xc[a:b] = ff[np.arange(b - a), im]
                fi[a:b] = im
            else:
                xc[a:b] = ff
        if flattenToTime:
            return xc, fi
        return xc


class IppGroupXcorrFFT:
    """Native twin of GroupXcorrFFT: always the full (S, fftlen) plane, float32.

    ref: cython_ext/CyGroupXcorrFFT/GroupXcorrFFT.cpp:3-203, CyGroupXcorrFFT.pyx:6-65.
    Differences from the Python class: complex64 arithmetic, group phases built
    in float64 then cast (:53-75), divisor cast to float32 before dividing (:142),
    offsets always zeroed by the first (:34), fftlen < groupLength is an error (:11-14).
    """

    def __init__(self, ygroups, offsets, fs, fftlen=-1, autoConj=True):
        ygroups = np.asarray(ygroups, dtype=np.complex64)
        offsets = np.asarray(offsets, dtype=np.int32)
        self.G, self.L = ygroups.shape
        self.fftlen = self.L if fftlen == -1 else int(fftlen)
        if self.fftlen < self.L:
            raise ValueError("fftlen must be >= group length")
        self.y = ygroups.conj() if autoConj else ygroups.copy()
        self.yNormSq = _energy(self.y.ravel())
        self.offsets = offsets - offsets[0]
        n = np.arange(self.fftlen, dtype=np.float64)
        self.phases = np.exp(-2j * np.pi * (self.offsets[:, None].astype(np.float64) * n[None, :]) / self.fftlen).astype(
            np.complex64
        )

    def xcorr(self, rx, shifts, NUM_THREADS=1):
        rx = np.asarray(rx, dtype=np.complex64)
        shifts = np.asarray(shifts, dtype=np.int32)
        out = np.zeros((shifts.size, self.fftlen), dtype=np.float32)
        rel = self.offsets[:, None] + np.arange(self.L)[None, :]
        for a, b in _chunks(shifts.size, self.G * self.fftlen):
            g = rx[shifts[a:b, None, None] + rel[None]]
            e = _energy(g.reshape(b - a, -1), 1)
            spec = _sfft.fft(g * self.y[None], n=self.fftlen, axis=2)
            comb = np.sum(spec * self.phases[None], axis=1).astype(np.complex64)
            p = comb.real**2 + comb.imag**2
            out[a:b] = p / (e * self.yNormSq).astype(np.float32)[:, None]
        return out


class TemplateCrossCorrelator:
    """T templates, no frequency scan, one M-point FFT/IFFT; returns QF (not QF^2).

    ref: xcorrRoutines.py:277-371 with the cupy calls replaced by NumPy.  The
    sliding norm is float32 (moving sum of float32 |x|^2 accumulated in double,
    filter.cu:291-347), zero-energy windows give inf/nan (Appendix B.10).
    The reference raises TypeError for non-device templates; the oracle takes
    host arrays (that check belongs to the product's device-array type).
    """

    def __init__(self, templates, inputSize):
        templates = np.asarray(templates)
        if templates.ndim != 2:
            raise ValueError("Templates must be a 2D array; 1 row for 1 template.")
        self._inputSize = int(inputSize)
        self._L = templates.shape[1]
        self._norms = np.sqrt(_energy(templates, 1)).astype(np.float32)
        padded = np.zeros((templates.shape[0], self._inputSize), dtype=templates.dtype)
        padded[:, : self._L] = templates
        self._tfc = _sfft.fft(padded, axis=1).conj()

    def correlate(self, x, returnMax=False):
        x = np.asarray(x)
        if x.ndim != 1 or x.size != self._inputSize:
            raise ValueError("x must be 1D of length %d" % self._inputSize)
        L = self._L
        xf = _sfft.fft(x)
        pw = (x.real.astype(np.float64) ** 2 + x.imag.astype(np.float64) ** 2).astype(np.float32)
        cs = np.concatenate(([0.0], np.cumsum(pw.astype(np.float64))))
        mov = (cs[L:] - cs[:-L]).astype(np.float32)  # == movingAverage(sumInstead)[L-1:]
        norms = mov**0.5
        out = _sfft.ifft(xf * self._tfc, axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            nout = out[:, : x.size - L + 1] / norms
            nout = nout / self._norms.reshape((-1, 1))
        if not returnMax:
            return nout
        a = np.abs(nout)
        ti = np.argmax(a, axis=0)
        return a[ti, np.arange(a.shape[1])], ti


class IppXcorrFFT:
    """Native threaded twin of fastXcorr branch B.

    ref: IppXcorrFFT.cpp:94-194, CyIppXcorrFFT.pyx:25-80.  float32 QF^2 with the
    divisor cast to float32 before dividing (:174,:193); int32 bin index (first
    maximum); out-of-range delays produce (0.0, 0) instead of an error (:125-130).
    """

    def __init__(self, cutout, num_threads=1, autoConj=True):
        cutout = np.asarray(cutout, dtype=np.complex64)
        self.c = cutout.conj() if autoConj else cutout.copy()
        self.n = cutout.size
        self.cNormSq = np.float32(_energy(cutout))

    def xcorr(self, rx, startIdx, endIdx, step):
        rx = np.asarray(rx, dtype=np.complex64)
        idx = np.arange(startIdx, endIdx, step)
        pk = np.zeros(idx.size, dtype=np.float32)
        fi = np.zeros(idx.size, dtype=np.int32)
        ok = (idx >= 0) & (idx + self.n <= rx.size)
        good = np.nonzero(ok)[0]
        for a, b in _chunks(good.size, self.n):
            sel = good[a:b]
            w = _windows(rx, self.n, idx[sel])
            spec = _sfft.fft(w * self.c, axis=1)
            p = spec.real**2 + spec.imag**2
            im = np.argmax(p, axis=1)
            e = _energy(w, 1)
            pk[sel] = p[np.arange(b - a), im] / self.cNormSq / e.astype(np.float32)
            fi[sel] = im
        return pk, fi


def cp_fastXcorr(cutout, rx, freqsearch=True, outputCAF=False, shifts=None, absResult=True, BATCH=1024, copyToCpu=True):
    """GPU v1 semantics (xcorrRoutines.py:29-167): only the freqsearch / no-CAF /
    abs branch exists; returns (float64 QF^2, uint32 bin)."""
    if not freqsearch or outputCAF or not absResult:
        print("Not implemented.")
        return None
    return fastXcorr(cutout, rx, freqsearch=True, shifts=shifts)


def cp_fastXcorr_v2(cutout, rx, startIdx=0, idxlen=None, cztObj=None, flattenCAF=False):
    """GPU v2 semantics (xcorrRoutines.py:169-274): float32 outputs, return order
    (freqIdx uint32, qf2 float32) -- swapped vs v1 (Appendix B.2) -- or the
    (idxlen, N | k) float32 plane.  The upstream multi-batch indexing bug in the
    non-flatten mode (:263-265) is not reproduced: every row is its own CAF row."""
    cutout = np.asarray(cutout, dtype=np.complex64)
    rx = np.asarray(rx, dtype=np.complex64)
    n = cutout.size
    if idxlen is None:
        idxlen = rx.size - n - startIdx + 1
    if cztObj is not None and cztObj.m != n:
        raise ValueError("CZT object input length doesn't match the cutout array size")
    from . import kernels as _k

    rows = _k.slidingMultiplyNormalised(cutout, rx, startIdx, idxlen)
    spec = _sfft.fft(rows, axis=1) if cztObj is None else cztObj.runMany(rows).astype(np.complex64)
    p = (spec.real.astype(np.float32) ** 2 + spec.imag.astype(np.float32) ** 2).astype(np.float32)
    if flattenCAF:
        return np.argmax(p, axis=1).astype(np.uint32), p.max(axis=1)
    return p
