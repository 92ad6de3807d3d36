"""Seeded random shapes through the kernel-level entry points that have tiled / register-tiled fast paths
(FIR undecimated and decimating, fused int16 front end, moving average, multi-template sliding dot product,
local maxima + gather), each against its NumPy / SciPy definition (filterRoutines.py:1245-1365,
benchmark_filterkernels.py:72-74, cupyExtensions.py:563-686).  CAF_FUZZ_CASES=N widens the seed range."""

import os

import numpy as np
import pytest
import scipy.signal as sps

import oracle as O
from oracle import kernels as K
from conftest import cn

pytestmark = pytest.mark.gpu
CASES = list(range(int(os.environ.get("CAF_FUZZ_CASES", "16"))))


@pytest.mark.parametrize("seed", CASES)
def test_fir_random_shapes(seed):
    from pydsproutines_amd import asarray
    from pydsproutines_amd.filterRoutines import CupyKernelFilter
    from pydsproutines_amd.usrpRoutines import Iq16FrontEnd

    rng = np.random.default_rng(5000 + seed)
    nt = int(rng.choice([1, 2, 7, 8, 9, 33, 64, 127, 128, 500, 1023, 2048, 2049, 3000]))
    n = int(rng.integers(1, 30000))
    dsr = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 11, 16, 17]))
    ph = int(rng.integers(0, dsr))
    taps = (rng.standard_normal(nt) / np.sqrt(nt)).astype(np.float32)
    x = cn(rng, n)
    ref = sps.lfilter(taps.astype(np.float64), 1, x.astype(np.complex128))[ph::dsr]
    got = CupyKernelFilter().filter_smtaps(asarray(x), asarray(taps), dsr=dsr, dsPhase=ph).get()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, atol=4e-5)
    if nt <= 2048 and dsr <= 16:
        raw = rng.integers(-3000, 3000, 2 * n, dtype=np.int16)
        xr = (raw.astype(np.float32) * np.float32(1 / 4096)).view(np.complex64)
        refr = sps.lfilter(taps.astype(np.float64), 1, xr.astype(np.complex128))[ph::dsr]
        fe = Iq16FrontEnd(asarray(taps), dsr, ph, 1 / 4096)
        cut = int(rng.integers(0, n + 1))
        parts = [fe.run(asarray(raw[2 * a : 2 * b])).get() for a, b in ((0, cut), (cut, n)) if b > a]
        np.testing.assert_allclose(np.concatenate(parts), refr, atol=4e-5)


@pytest.mark.parametrize("seed", CASES)
def test_moving_average_random_shapes(seed):
    from pydsproutines_amd import asarray
    from pydsproutines_amd.filterRoutines import cupyMovingAverage, cupyMultiMovingAverage

    rng = np.random.default_rng(6000 + seed)
    n = int(rng.integers(1, 40000))
    L = int(rng.choice([1, 2, 31, 100, 1000, 1024, 1025, 4095, 4096, 5000]))
    x = rng.standard_normal(n).astype(np.float32)
    np.testing.assert_allclose(cupyMovingAverage(asarray(x), L).get(), K.movingAverage(x, L), rtol=1e-6, atol=3e-6)
    np.testing.assert_allclose(cupyMovingAverage(asarray(x), L, sumInstead=True).get(), K.movingAverage(x, L, True),
                               rtol=1e-6, atol=1e-4)
    rows = int(rng.integers(1, 6))
    x2 = rng.standard_normal((rows, max(2, n // rows))).astype(np.float32)
    np.testing.assert_allclose(cupyMultiMovingAverage(asarray(x2), L).get(), K.multiMovingAverage(x2, L), rtol=1e-6,
                               atol=3e-6)


@pytest.mark.parametrize("seed", CASES)
def test_multi_template_dot_random_shapes(seed):
    from pydsproutines_amd import asarray
    from pydsproutines_amd.cupyExtensions import multiTemplateSlidingDotProduct

    rng = np.random.default_rng(7000 + seed)
    T = int(rng.integers(1, 9))
    L = int(rng.choice([1, 5, 8, 16, 17, 100, 255, 256, 1000, 2048, 2055]))
    nx = int(rng.integers(L + 1, L + 12000))
    st = int(rng.integers(0, max(1, (nx - L) // 3)))
    nsl = int(rng.integers(1, nx - L + 1 - st + 1))
    tm = cn(rng, T * L).reshape(T, L)
    x = cn(rng, nx)
    p = st + nsl // 2
    x[p : p + L] += 5 * tm[T - 1].conj()
    ti, q = multiTemplateSlidingDotProduct(asarray(x), asarray(tm), st, nsl)
    oti, oq = K.multiTemplateSlidingDotProduct(x, tm, st, nsl)
    np.testing.assert_allclose(q.get(), oq, atol=2e-5)
    if L > 1:  # (one-sample templates all score exactly 1: the index is decided by rounding)
        clear = oq > 0.6
        np.testing.assert_array_equal(ti.get()[clear], oti[clear])
        assert clear[nsl // 2] and oti[nsl // 2] == T - 1


@pytest.mark.parametrize("seed", CASES)
def test_local_maxima_random_shapes(seed):
    from pydsproutines_amd import asarray
    from pydsproutines_amd.cupyExtensions import cupyFindLocalMaxima
    from pydsproutines_amd.zoom import gather, topk_local_maxima

    rng = np.random.default_rng(8000 + seed)
    n = int(rng.choice([1, 2, 3, 4095, 4096, 4097, int(rng.integers(5, 300000))]))
    v = np.abs(rng.standard_normal(n)).astype(np.float32)
    if rng.integers(0, 2):
        v[rng.integers(0, n, max(1, n // 7))] = 0.0  # plateaus and repeated values
    h = float(rng.choice([0.0, 0.5, 2.0, 10.0]))
    ref = K.findLocalMaxima(v, h)
    dv = asarray(v)
    idx, cnt = cupyFindLocalMaxima(dv, h, maxNumPeaks=max(1, ref.size))
    assert int(cnt.get()[0]) == ref.size
    np.testing.assert_array_equal(idx.get()[: ref.size], ref)
    np.testing.assert_array_equal(gather(dv, idx, ref.size), v[ref])
    k = int(rng.integers(1, 9))
    ti, tv = topk_local_maxima(dv, k, h, maxNumPeaks=max(1, ref.size))
    np.testing.assert_array_equal(ti, K.topk_peaks(v, h, k))
    np.testing.assert_array_equal(tv, v[ti])


@pytest.mark.parametrize("seed", CASES)
def test_elementwise_and_resampling_random_shapes(seed):
    """upfirdn (naive and shared-memory forms), complex moving sum, sliding normalised product, row argmax,
    |.|^2, slice copies, tone dot products (upfirdn.cu, filter.cu:374-438, multiplySlices.cu:113-216,
    argmax.cu:93-153, complex_magn.cu, copying.cu, genTones.cu:165-283)."""
    from pydsproutines_amd import asarray
    from pydsproutines_amd.cupyExtensions import (
        cupyArgmaxAbsRows_complex64,
        cupyComplexMagnSq,
        cupyCopyEqualSlicesToMatrix_32fc,
        cupyCopyIncrementalEqualSlicesToMatrix_32fc,
        multiplySlidesNormalised,
    )
    from pydsproutines_amd.filterRoutines import CupyKernelFilter, cupyComplexMovingSum
    from pydsproutines_amd.spectralRoutines import cupyDotTonesScaling

    rng = np.random.default_rng(8500 + seed)
    f = CupyKernelFilter()
    # upfirdn
    up, down = int(rng.integers(1, 12)), int(rng.integers(1, 9))
    nt = int(rng.choice([1, 3, 32, 64, 129, 500]))
    taps = (rng.standard_normal(nt) / np.sqrt(nt)).astype(np.float32)
    rows, n = int(rng.integers(1, 5)), int(rng.integers(1, 3000))
    xm = cn(rng, rows * n).reshape(rows, n)
    ref = sps.upfirdn(taps.astype(np.float64), xm.astype(np.complex128), up, down, axis=1)
    got = f.upfirdn_sm(asarray(xm), asarray(taps), up, down).get()
    assert got.shape == ref.shape and ref.shape[1] == f.getUpfirdnSize(n, nt, up, down)
    np.testing.assert_allclose(got, ref, atol=4e-5)
    g1 = f.upfirdn_naive(asarray(xm[0].copy()), asarray(taps), up, down).get()
    np.testing.assert_allclose(g1, ref[0], atol=4e-5)
    # complex moving sum
    n2 = int(rng.integers(2, 20000))
    L = int(rng.integers(1, min(n2, 700) + 1))
    z = cn(rng, n2)
    got = cupyComplexMovingSum(asarray(z), L).get()
    np.testing.assert_allclose(got, K.movingComplexSum(z, L), rtol=3e-4, atol=1e-4)
    # sliding normalised product + row argmax + |.|^2
    xl, yl = int(rng.integers(1, 300)), int(rng.integers(301, 5000))
    x, y = cn(rng, xl), cn(rng, yl)
    start = int(rng.integers(0, yl - xl))
    cnt = int(rng.integers(1, min(400, yl - start) + 1))
    coef = float(rng.uniform(0.5, 3.0)) if rng.integers(0, 2) else None
    dz = multiplySlidesNormalised(asarray(x), asarray(y), start, cnt, coefficient=None if coef is None else np.array([coef]))
    oz = K.slidingMultiplyNormalised(x, y, start, cnt, coef)
    np.testing.assert_allclose(dz.get(), oz, atol=3e-6)
    am, mx = cupyArgmaxAbsRows_complex64(dz, returnMaxValues=True)
    oam, omx = K.argmaxAbsRows(dz.get())
    np.testing.assert_array_equal(am.get(), oam)
    np.testing.assert_allclose(mx.get(), omx, rtol=1e-6)
    np.testing.assert_allclose(cupyComplexMagnSq(dz, np.float32).get(), K.complexMagnSq(dz.get(), np.float32), rtol=1e-6)
    # slice copies
    ln = int(rng.integers(1, 200))
    stv = rng.integers(0, yl - ln, int(rng.integers(1, 30))).astype(np.int32)
    np.testing.assert_array_equal(cupyCopyEqualSlicesToMatrix_32fc(asarray(y), asarray(stv), ln).get(),
                                  K.copySlicesToMatrix(y, stv, ln))
    inc, nr = int(rng.integers(1, 20)), int(rng.integers(1, 40))
    s0 = int(rng.integers(0, max(1, yl - ln - inc * nr)))
    if s0 + inc * (nr - 1) + ln <= yl:
        np.testing.assert_array_equal(cupyCopyIncrementalEqualSlicesToMatrix_32fc(asarray(y), s0, inc, ln, nr).get(),
                                      K.copyIncrementalEqualSlicesToMatrix(y, s0, inc, ln, nr))
    # tone dot products
    nf = int(rng.integers(1, 80))
    src = cn(rng, int(rng.integers(1, 6000)))
    f0, fstep = float(rng.uniform(-0.4, 0.4)), float(rng.uniform(1e-5, 1e-2))
    got = cupyDotTonesScaling(f0, fstep, nf, asarray(src)).get()
    np.testing.assert_allclose(got, K.dotTonesScaling(f0, fstep, nf, src), atol=3e-5 * np.sqrt(src.size) * 4)


def _smooth_lengths(limit, max_prime=23):
    out = []
    for n in range(32, limit + 1):
        r = n
        for p in (2, 3, 5, 7, 11, 13, 17, 19, 23):
            if p > max_prime:
                break
            while r % p == 0:
                r //= p
        if r == 1:
            out.append(n)
    return out


_LENGTHS = _smooth_lengths(16384)
_LONG_LENGTHS = [n for n in _smooth_lengths(120000) if n > 16384]  # up to 20000: one LDS image; beyond: residues (caf_perdelay_jit.h, PDJ_Q)


@pytest.mark.parametrize("seed", CASES)
def test_perdelay_random_cutout_lengths(seed):
    """caf_xcorr_perdelay at random cutout lengths whose prime factors are at most 23 -- whatever kernel the library routes them to
    (powers of two / ten, or the kernel it compiles for the length at run time with the plan its model picks) -- against the
    oracle's fastXcorr(freqsearch=True): maxima to 2e-5, bins exact where the oracle's top-2 margin is clear, planes, strides in
    both directions, the (0, 0) rule for windows that leave rx.  Every other seed adds a length of 16385 ... 120000 samples, every
    seed an arbitrary length up to 10000 (whatever its factors)."""
    import ctypes as ct

    from pydsproutines_amd import _lib, asarray
    from pydsproutines_amd.devarray import empty

    rng = np.random.default_rng(9100 + seed)
    lib = _lib.load()
    # two lengths per seed: one anywhere, one short (many rows per workgroup)
    lens = [int(rng.choice(_LENGTHS)), int(rng.choice([v for v in _LENGTHS if v <= 600]))]
    if seed % 2 == 0:
        lens.append(int(rng.choice(_LONG_LENGTHS)))
    lens.append(int(rng.integers(16, 10001)))  # any length at all: mostly a prime factor above 23 (Bluestein in LDS up to 10000 samples)
    for n in lens:
        num = int(rng.integers(3, 12 if n > 16384 else (40 if n > 4000 else 120)))
        step = int(rng.choice([1, 1, 2, -1, 5]))
        m = n + abs(step) * num + int(rng.integers(0, 50))
        rx = cn(rng, m)
        start = 0 if step > 0 else m - n
        start += int(rng.integers(0, 3)) * (1 if step > 0 else -1)
        sh = start + step * np.arange(num)
        sh = sh[(sh >= 0) & (sh + n <= m)]
        num = sh.size
        k0 = int(rng.integers(0, n))
        cut = (rx[sh[num // 2] : sh[num // 2] + n] * np.exp(-2j * np.pi * k0 * np.arange(n) / n)).astype(np.complex64)
        rx = (rx + 0.1 * cn(rng, m)).astype(np.complex64)
        d_rx, d_cut = asarray(rx), asarray(cut.conj().copy())
        q, fi, pl = empty(num, np.float32), empty(num, np.int32), empty((num, n), np.float32)
        p = lambda a: ct.c_void_p(a.ptr)  # noqa: E731
        want_plane = bool(rng.integers(0, 2))
        _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), m, int(sh[0]), step, num, 0, p(q), p(fi), p(pl) if want_plane else None,
                                          None, 0, None))
        rq, rf = O.fastXcorr(cut, rx, freqsearch=True, shifts=sh)
        gq, gf = q.get(), fi.get()
        assert np.max(np.abs(gq - rq)) <= 2e-5, (n, step)
        assert (int(np.argmax(gq)), int(gf[np.argmax(gq)])) == (num // 2, k0), (n, step)
        diff = np.nonzero(gf != rf)[0]
        if diff.size:  # float32 ties only
            rows = O.fastXcorr(cut, rx, freqsearch=True, outputCAF=True, shifts=sh[diff])
            assert np.all(rows[np.arange(diff.size), gf[diff]] >= rows.max(axis=1) - 4e-5), (n, step)
        if want_plane:
            gp = pl.get()
            np.testing.assert_array_equal(gq, gp.max(axis=1))
            np.testing.assert_array_equal(gf, np.argmax(gp, axis=1))
