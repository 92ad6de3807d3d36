"""Seeded random shapes through the kernel-level entry points that have tiled / register-tiled fast paths
(FIR undecimated and decimating, fused int16 front end, moving average, multi-template sliding dot product,
local maxima + gather), each against its NumPy / SciPy definition (filterRoutines.py:1245-1365,
benchmark_filterkernels.py:72-74, cupyExtensions.py:563-686).  CAF_FUZZ_CASES=N widens the seed range."""

import os

import numpy as np
import pytest
import scipy.signal as sps

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
