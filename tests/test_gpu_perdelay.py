"""GPU parity tests of the fused per-delay correlator (csrc/caf_perdelay.hip) behind caf_xcorr_perdelay:
every power-of-two cutout length 64 ... 16384 against the oracle's fastXcorr branches B / C / C'
(the reference's literal algorithm, xcorrRoutines.py:511-580), strided and descending delay runs, the
out-of-range rules of both native twins (IppXcorrFFT.cpp:125-130 zero rows; multiplySlices.cu:147-163 zero
padding), and agreement with the three-kernel form it replaces (CAF_PERDELAY_UNFUSED=1, separate process)."""

import ctypes as ct
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
from conftest import REPO, cn

pytestmark = pytest.mark.gpu


def _perdelay(cut_conj, rx, start, step, num, zero_oor=False, caf=False, ccaf=False):
    from pydsproutines_amd import _lib, asarray
    from pydsproutines_amd.devarray import empty

    lib = _lib.load()
    n = cut_conj.size
    d_cut, d_rx = asarray(cut_conj), asarray(rx)
    qf2, idx = empty(num, np.float32), empty(num, np.int32)
    pl = empty((num, n), np.float32) if caf else None
    cp = empty((num, n), np.complex64) if ccaf else None
    p = lambda a: ct.c_void_p(a.ptr) if a is not None else None  # noqa: E731
    _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), rx.size, int(start), int(step), int(num), 1 if zero_oor else 0,
                                      p(qf2), p(idx), p(pl), p(cp), 0, None), "caf_xcorr_perdelay")
    _lib.check(lib.caf_stream_sync(None))
    return qf2.get(), idx.get(), (pl.get() if caf else None), (cp.get() if ccaf else None)


@pytest.mark.parametrize("log2n", range(6, 15))
def test_fused_perdelay_matches_oracle(log2n):
    n = 1 << log2n
    rng = np.random.default_rng(100 + log2n)
    m = n + 700
    rx = cn(rng, m)
    d0, k0 = 123, (3 * n) // 8 + 1
    cut = (rx[d0 : d0 + n] * np.exp(-2j * np.pi * k0 * np.arange(n) / n)).astype(np.complex64)  # planted at (d0, k0)
    rx = (rx + 0.05 * cn(rng, m)).astype(np.complex64)
    num = 600 if n <= 4096 else 150
    shifts = np.arange(num)
    q, fi, pl, cp = _perdelay(cut.conj(), rx, 0, 1, num, caf=True, ccaf=True)
    ref_c = O.fastXcorr(cut, rx, freqsearch=True, outputCAF=True, shifts=shifts)                  # branch C
    ref_cc = O.fastXcorr(cut, rx, freqsearch=True, outputCAF=True, shifts=shifts, absResult=False)  # branch C'
    tol = 2e-5
    assert np.max(np.abs(pl - ref_c)) <= tol
    assert np.max(np.abs(cp - ref_cc)) <= 1e-4 * max(1.0, np.abs(ref_cc).max())
    np.testing.assert_array_equal(q, pl.max(axis=1))              # row results == the plane it wrote
    np.testing.assert_array_equal(fi, np.argmax(pl, axis=1))      # first index of the maximum
    assert (int(np.argmax(q)), int(fi[np.argmax(q)])) == (d0, k0)
    top2 = np.sort(ref_c, axis=1)[:, -2:]
    clear = top2[:, 1] - top2[:, 0] > 4 * tol
    np.testing.assert_array_equal(fi[clear], np.argmax(ref_c, axis=1)[clear])


def test_fused_perdelay_strides_and_out_of_range_rules():
    n = 256
    rng = np.random.default_rng(7)
    rx = cn(rng, 3000)
    cut = cn(rng, n)
    # strided ascending and descending runs
    for start, step, num in ((5, 7, 300), (2700, -9, 280), (0, 1, 3000 - n + 1)):
        q, fi, _, _ = _perdelay(cut.conj(), rx, start, step, num)
        sh = start + step * np.arange(num)
        rq, rf = O.fastXcorr(cut, rx, freqsearch=True, shifts=sh)
        assert np.max(np.abs(q - rq)) <= 2e-5
        # the argmax rule: where the reported bin is not the oracle's, the oracle's own row holds a value
        # within 2 tol of its maximum at the reported bin -- a float32 tie, not a different peak
        diff = np.nonzero(fi != rf)[0]
        if diff.size:
            rows = O.fastXcorr(cut, rx, freqsearch=True, outputCAF=True, shifts=sh[diff])
            assert np.all(rows[np.arange(diff.size), fi[diff]] >= rows.max(axis=1) - 2 * 2e-5)
    # CyIppXcorrFFT rule: windows that leave rx give (0, 0)
    q, fi, pl, _ = _perdelay(cut.conj(), rx, -40, 1, 3000 - n + 90, zero_oor=True, caf=True)
    inside = (np.arange(-40, 3000 - n + 50) >= 0) & (np.arange(-40, 3000 - n + 50) + n <= 3000)
    assert np.all(q[~inside] == 0) and np.all(fi[~inside] == 0) and np.all(pl[~inside] == 0)
    rq, rf = O.fastXcorr(cut, rx, freqsearch=True, shifts=np.arange(0, 3000 - n + 1))
    assert np.max(np.abs(q[inside] - rq)) <= 2e-5
    # an all-zero window: NaN plane, (NaN, 0) row result -- what the reference's `pmax / cutoutNormSq / 0` gives
    # (xcorrRoutines.py:527-528): the oracle itself is asked
    rz = rx.copy()
    rz[1000 : 1000 + n + 10] = 0
    q, fi, pl, _ = _perdelay(cut.conj(), rz, 1000, 1, 8, caf=True)
    assert np.all(np.isnan(q)) and np.all(fi == 0) and np.all(np.isnan(pl))
    with np.errstate(all="ignore"):
        rq0, rf0 = O.fastXcorr(cut, rz, freqsearch=True, shifts=np.arange(1000, 1008))
    assert np.all(np.isnan(rq0)) and np.all(rf0 == 0)


def test_4096_rules_and_long_runs():
    """N = 4096 (one row per workgroup and turn): odd row counts, strides, the out-of-range rule at both ends, an all-zero
    window, and a run long enough for several rows per workgroup (rows_per_wg = 4)."""
    n = 4096
    rng = np.random.default_rng(17)
    rx = cn(rng, 30000)
    cut = cn(rng, n)
    tol = 2e-5

    def check(q, fi, sh, zero_rule=False):
        inside = (sh >= 0) & (sh + n <= rx.size)
        if zero_rule:
            assert np.all(q[~inside] == 0) and np.all(fi[~inside] == 0)
            q, fi, sh = q[inside], fi[inside], sh[inside]
        rq, rf = O.fastXcorr(cut, rx, freqsearch=True, shifts=sh)
        assert np.max(np.abs(q - rq)) <= tol
        diff = np.nonzero(fi != rf)[0]
        if diff.size:  # float32 ties only
            rows = O.fastXcorr(cut, rx, freqsearch=True, outputCAF=True, shifts=sh[diff])
            assert np.all(rows[np.arange(diff.size), fi[diff]] >= rows.max(axis=1) - 2 * tol)

    for start, step, num in ((3, 1, 301), (11, 3, 77), (25000, -7, 501), (0, 1, 17001)):
        q, fi, _, _ = _perdelay(cut.conj(), rx, start, step, num)
        check(q, fi, start + step * np.arange(num))
    # planes of an odd run: the row results are the planes' own maxima, bit for bit
    q, fi, pl, cp = _perdelay(cut.conj(), rx, 100, 1, 33, caf=True, ccaf=True)
    np.testing.assert_array_equal(q, pl.max(axis=1))
    np.testing.assert_array_equal(fi, np.argmax(pl, axis=1))
    ref = O.fastXcorr(cut, rx, freqsearch=True, outputCAF=True, shifts=np.arange(100, 133), absResult=False)
    assert np.max(np.abs(cp - ref)) <= 1e-4 * max(1.0, np.abs(ref).max())
    # CyIppXcorrFFT rule: zero rows outside rx (IppXcorrFFT.cpp:125-130), at both ends
    sh = np.arange(-20, -20 + 151)
    q, fi, _, _ = _perdelay(cut.conj(), rx, -20, 1, 151, zero_oor=True)
    check(q, fi, sh, zero_rule=True)
    sh = np.arange(rx.size - n - 30, rx.size - n + 31)
    q, fi, _, _ = _perdelay(cut.conj(), rx, int(sh[0]), 1, sh.size, zero_oor=True)
    check(q, fi, sh, zero_rule=True)
    with pytest.raises(ValueError):  # without the rule such a run is refused (include/caf.h)
        _perdelay(cut.conj(), rx, rx.size - n - 3, 1, 8)
    # an all-zero window inside a sliding run: NaN plane, (NaN, 0) row results, and the rows after it are right again
    rz = rx.copy()
    rz[9000 : 9000 + n + 5] = 0
    q, fi, pl, _ = _perdelay(cut.conj(), rz, 8990, 1, 40, caf=True)
    dead = (np.arange(8990, 9030) >= 9000) & (np.arange(8990, 9030) <= 9005)
    assert np.all(np.isnan(q[dead])) and np.all(fi[dead] == 0) and np.all(np.isnan(pl[dead]))
    rq, _ = O.fastXcorr(cut, rz, freqsearch=True, shifts=np.arange(8990, 9030)[~dead])
    assert np.max(np.abs(q[~dead] - rq)) <= tol


@pytest.mark.parametrize("n", [100, 1000, 10000])
def test_decimal_cutouts_radix10_kernel(n):
    """Cutouts of 100 / 1000 / 10000 samples (benchmark_xcorrs.py's default is 1000) run the same per-delay algorithm with
    radix-10 passes in LDS (k_perdelay_r10): the oracle's branches B, C, C' again, a planted (delay, bin), odd row counts,
    strides in both directions, the zero rows of the out-of-range rule and an all-zero window."""
    _composite_cutout_checks(n)


# 2^a 3^b 5^c lengths that are neither a power of two nor of ten, with the planner's own plans: short ones (several rows per
# workgroup) and long ones (one row, both launch-bound variants beyond 8192 samples), odd ones (243 = 3^5, 75 = 5.5.3), the
# benchmark's kind of lengths (1200, 1536, 3000, 5000, 12000); forced plans: test_mixed_radix_every_butterfly
@pytest.mark.parametrize("n", [36, 48, 60, 72, 75, 243, 486, 1200, 1536, 3000, 5000, 12000, 15552, 16200])
def test_mixed_radix_cutouts(n):
    """Cutouts of 2^a 3^b 5^c samples run the per-delay algorithm in ONE kernel too (k_perdelay_mr, caf_perdelay_mr.hip: a
    mixed-radix Stockham transform in LDS; the planner picks radices and threads per row) instead of product rows -> rocFFT rows ->
    argmax through HBM: the same checks as the other two kernels against the oracle's branches B, C, C'
    (xcorrRoutines.py:511-566)."""
    _composite_cutout_checks(n)


# every butterfly of the mixed-radix kernel as the first pass (the kernel instances) and as a later pass, through forced plans
# (CAF_MR_PLAN = radices / threads per row; the planner would pick other ones for some of these lengths)
@pytest.mark.parametrize("n,plan", [(125, "5,5,5/8"), (49, "7,7/4"), (343, "7,7,7/25"), (72, "8,3,3/5"), (81, "9,9/6"), (50, "10,5/4"),
                                    (36, "12,3/3"), (98, "14,7/7"), (225, "15,15/15"), (1280, "16,16,5/80"), (324, "18,9,2/21"),
                                    (400, "20,20/25"), (8000, "20,20,20/500"), (96, "16,6/6"), (80, "20,4/5"), (40, "20,2/3"),
                                    (1200, "15,10,8/80"), (1200, "16,5,5,3/75"), (1400, None), (7000, None), (2401, None),
                                    (14000, None), (12005, None)])
def test_mixed_radix_every_butterfly(n, plan, monkeypatch, capfd):
    """Radices 2 ... 10, 12, 14, 15, 16, 18, 20 (the composite ones as Cooley-Tukey butterflies in registers, radix 7 for
    7-smooth lengths such as 1400 = 2^3 5^2 7), first and later passes, against the oracle as above."""
    monkeypatch.setenv("CAF_MR_DEBUG", "1")
    monkeypatch.setenv("CAF_JIT", "0")  # the prebuilt, plan-driven kernel (what a box without hiprtc runs)
    if plan:
        monkeypatch.setenv("CAF_MR_PLAN", plan)
    _composite_cutout_checks(n)
    err = capfd.readouterr().err
    assert "[caf mr] n=%d plan=" % n in err
    if plan:
        assert "[caf mr] n=%d plan=%s " % (n, plan) in err  # (the forced plan was valid and is the one that ran)


# the same lengths and butterflies through the kernel compiled for the length at run time (caf_jit.hip, caf_perdelay_jit.h: in-place
# decimation in frequency, layout by bank simulation); forced plans put every radix first, in the middle and last
@pytest.mark.parametrize("n,plan", [(1200, None), (1400, None), (5000, None), (1536, None), (3000, None), (96, None), (12000, None),
                                    (16200, None), (14336, None), (360, None), (15000, None), (48, None), (2401, None),
                                    (125, "5,5,5/8"), (343, "7,7,7/25"), (72, "3,3,8/5"), (81, "9,9/6"), (50, "5,10/4"), (36, "3,12/3"),
                                    (98, "7,14/7"), (225, "15,15/15"), (1280, "5,16,16/80"), (324, "2,9,18/21"), (400, "20,20/25"),
                                    (8000, "20,20,20/500"), (96, "6,16/6"), (1200, "8,10,15/80"), (1200, "3,5,5,16/75"),
                                    (1250, "25,25,2/64"), (2500, "4,25,25/125"), (7000, "7,10,10,10/500"), (7776, "6,6,6,6,6/432"),
                                    # prime factors 11 .. 23 (direct-form butterflies): lengths rounds 1-4 sent through rocFFT rows
                                    (1430, None), (1001, None), (2431, None), (46, None), (33, None), (4199, None), (253, "11,23/23"),
                                    (2873, "13,17,13/221"), (361, "19,19/19"),
                                    # beyond 16384 samples: as long as a row image fits the 160 KB of LDS
                                    (20000, None), (18000, None)])
def test_jit_kernel_lengths_and_butterflies(n, plan, monkeypatch, capfd):
    monkeypatch.setenv("CAF_JIT_DEBUG", "1")
    monkeypatch.delenv("CAF_JIT", raising=False)
    if plan:
        monkeypatch.setenv("CAF_PDJ_PLAN", plan)
    _composite_cutout_checks(n)
    err = capfd.readouterr().err
    assert "[caf jit] n=%d plan=" % n in err, err
    if plan:
        assert "[caf jit] n=%d plan=%s " % (n, plan) in err  # (the forced plan was valid and is the one that ran)


# cutouts longer than one LDS image: Q residues of an (n / Q)-point transform per row (caf_perdelay_jit.h, PDJ_Q) -- forced on short
# lengths for every Q, then the lengths that need it (rounds 1-4: product rows -> rocFFT rows -> argmax through HBM)
@pytest.mark.parametrize("n,q", [(2400, 2), (3600, 3), (4800, 4), (6000, 5), (7200, 6), (8400, 7), (9600, 8), (2002, 2), (96, 3), (5120, 16), (3744, 13),
                                 (19200, None), (24000, None), (32768, None), (40000, None), (50000, None), (65536, None), (100000, None)])
def test_jit_kernel_split_form(n, q, monkeypatch, capfd):
    monkeypatch.setenv("CAF_JIT_DEBUG", "1")
    monkeypatch.delenv("CAF_JIT", raising=False)
    if q:
        monkeypatch.setenv("CAF_PDJ_Q", str(q))
        monkeypatch.setenv("CAF_JIT_ALL", "1")
    _composite_cutout_checks(n)
    err = capfd.readouterr().err
    assert "[caf jit] n=%d plan=" % n in err, err
    if q:
        assert " residues=%d " % q in err
    else:
        assert " residues=1 " not in err


# cutout lengths with a prime factor above 23 -- most lengths a burst happens to have -- as Bluestein's chirp transform in the same
# LDS image (caf_perdelay_jit.h, PDJ_BLU): forward transform of the chirped products, product with the transformed chirp, the
# transposed passes back.  Natural lengths (primes, 2 x prime, the longest that fits), smooth lengths forced that way, forced
# convolution lengths with radices 3 / 5 / 7 in them.
@pytest.mark.parametrize("n,m", [(1450, None), (1021, None), (2047, None), (4099, None), (9973, None), (58, None), (37, None), (6001, None),
                                 (211, None), (1200, 0), (4096, 0), (100, 200), (1450, 2916), (1000, 2401), (997, 2000), (3001, 6075)])
def test_jit_kernel_bluestein(n, m, monkeypatch, capfd):
    monkeypatch.setenv("CAF_JIT_DEBUG", "1")
    monkeypatch.delenv("CAF_JIT", raising=False)
    if m is not None:
        monkeypatch.setenv("CAF_PDJ_BLUESTEIN", "1")
        monkeypatch.setenv("CAF_JIT_ALL", "1")
    if m:
        monkeypatch.setenv("CAF_PDJ_BLU_M", str(m))
    _composite_cutout_checks(n)
    err = capfd.readouterr().err
    assert "[caf jit] n=%d plan=" % n in err, err
    assert " bluestein=0 " not in err
    if m:
        assert " bluestein=%d " % m in err


def test_jit_and_prebuilt_kernels_agree(monkeypatch):
    """The two mixed-radix kernels (run-time compiled / plan-driven) on the same rows: same maxima to float32 rounding, same bins."""
    rng = np.random.default_rng(12)
    n, m = 1200, 30_000
    rx, cut = cn(rng, m), cn(rng, n)
    out = {}
    for jit in ("1", "0"):
        monkeypatch.setenv("CAF_JIT", jit)
        out[jit] = _perdelay(cut, rx, 3, 5, 4000, caf=True)
    assert np.max(np.abs(out["1"][0] - out["0"][0])) <= 2e-6
    np.testing.assert_allclose(out["1"][2], out["0"][2], atol=2e-6)
    same = out["1"][1] == out["0"][1]
    assert same.mean() > 0.999  # (float32 ties between two bins may fall either way)


def _composite_cutout_checks(n):
    rng = np.random.default_rng(n)
    m = n + 900
    rx = cn(rng, m)
    d0, k0 = 77, (3 * n) // 10 + 1
    cut = (rx[d0 : d0 + n] * np.exp(-2j * np.pi * k0 * np.arange(n) / n)).astype(np.complex64)  # planted at (d0, k0)
    rx = (rx + 0.05 * cn(rng, m)).astype(np.complex64)
    num = 301 if n <= 1000 else 101
    shifts = np.arange(num)
    q, fi, pl, cp = _perdelay(cut.conj(), rx, 0, 1, num, caf=True, ccaf=True)
    ref_c = O.fastXcorr(cut, rx, freqsearch=True, outputCAF=True, shifts=shifts)
    ref_cc = O.fastXcorr(cut, rx, freqsearch=True, outputCAF=True, shifts=shifts, absResult=False)
    tol = 2e-5
    assert np.max(np.abs(pl - ref_c)) <= tol
    assert np.max(np.abs(cp - ref_cc)) <= 1e-4 * max(1.0, np.abs(ref_cc).max())
    np.testing.assert_array_equal(q, pl.max(axis=1))
    np.testing.assert_array_equal(fi, np.argmax(pl, axis=1))
    assert (int(np.argmax(q)), int(fi[np.argmax(q)])) == (d0, k0)
    for start, step, cnt in ((5, 7, 111), (m - n, -3, 200)):
        q, fi, _, _ = _perdelay(cut.conj(), rx, start, step, cnt)
        sh = start + step * np.arange(cnt)
        rq, rf = O.fastXcorr(cut, rx, freqsearch=True, shifts=sh)
        assert np.max(np.abs(q - rq)) <= tol
        diff = np.nonzero(fi != rf)[0]
        if diff.size:  # float32 ties only
            rows = O.fastXcorr(cut, rx, freqsearch=True, outputCAF=True, shifts=sh[diff])
            assert np.all(rows[np.arange(diff.size), fi[diff]] >= rows.max(axis=1) - 2 * tol)
    # CyIppXcorrFFT rule: windows that leave rx give (0, 0) and zero rows
    q, fi, pl, _ = _perdelay(cut.conj(), rx, -13, 1, m - n + 40, zero_oor=True, caf=True)
    inside = (np.arange(-13, m - n + 27) >= 0) & (np.arange(-13, m - n + 27) + n <= m)
    assert np.all(q[~inside] == 0) and np.all(fi[~inside] == 0) and np.all(pl[~inside] == 0)
    rq, _ = O.fastXcorr(cut, rx, freqsearch=True, shifts=np.arange(0, m - n + 1))
    assert np.max(np.abs(q[inside] - rq)) <= tol
    # an all-zero window: NaN plane, (NaN, 0) row result
    rz = np.concatenate([rx, np.zeros(n + 10, np.complex64), rx[:50]])
    q, fi, pl, _ = _perdelay(cut.conj(), rz, m, 1, 8, caf=True)
    assert np.all(np.isnan(q)) and np.all(fi == 0) and np.all(np.isnan(pl))


@pytest.mark.parametrize("n", [10007, 20011])
def test_rows_path_for_lengths_no_kernel_takes(n):
    """Primes beyond the Bluestein image (10007) and beyond one LDS image (20011): product rows (k_sliding_multiply in its long-row
    form: four rows per workgroup, the row groups as the fast grid dimension) -> rocFFT rows -> argmax, against the oracle's
    branch B for consecutive, strided and descending delays and with the zero rows of the out-of-range rule."""
    from pydsproutines_amd import _lib

    assert not _lib.load().caf_xcorr_perdelay_one_kernel(n)
    rng = np.random.default_rng(n)
    m = n + 400
    rx = cn(rng, m)
    d0, k0 = 37, n // 3
    cut = (rx[d0 : d0 + n] * np.exp(-2j * np.pi * k0 * np.arange(n) / n)).astype(np.complex64)
    rx = (rx + 0.05 * cn(rng, m)).astype(np.complex64)
    tol = 2e-5
    for start, step, num in ((0, 1, 61), (5, 3, 23), (390, -7, 50), (30, 1, 3)):
        q, fi, _, _ = _perdelay(cut.conj(), rx, start, step, num)
        sh = start + step * np.arange(num)
        rq, rf = O.fastXcorr(cut, rx, freqsearch=True, shifts=sh)
        assert np.max(np.abs(q - rq)) <= tol
        diff = np.nonzero(fi != rf)[0]
        if diff.size:  # float32 ties only
            rows = O.fastXcorr(cut, rx, freqsearch=True, outputCAF=True, shifts=sh[diff])
            assert np.all(rows[np.arange(diff.size), fi[diff]] >= rows.max(axis=1) - 2 * tol)
        if start == 0:
            assert (int(np.argmax(q)), int(fi[np.argmax(q)])) == (d0, k0)
    sh = np.arange(-6, 12)
    q, fi, _, _ = _perdelay(cut.conj(), rx, -6, 1, sh.size, zero_oor=True)
    assert np.all(q[sh < 0] == 0) and np.all(fi[sh < 0] == 0)
    rq, _ = O.fastXcorr(cut, rx, freqsearch=True, shifts=sh[sh >= 0])
    assert np.max(np.abs(q[sh >= 0] - rq)) <= tol


def test_fused_equals_three_kernel_form():
    """The same calls through the path it replaces (rocFFT rows), in a child process because the switch is read once."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
from test_gpu_perdelay import _perdelay
from conftest import cn
rng = np.random.default_rng(5)
out = {}
for n in (64, 1000, 1024, 1200, 4096):
    rx = cn(rng, n + 500); cut = cn(rng, n)
    q, fi, pl, _ = _perdelay(cut.conj(), rx, 3, 2, 200, caf=True)
    out['q%%d' %% n], out['f%%d' %% n], out['p%%d' %% n] = q, fi, pl
np.savez(sys.argv[1], **out)
""" % (REPO, REPO)
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        res = {}
        for tag, env in (("fused", {}), ("unfused", {"CAF_PERDELAY_UNFUSED": "1"})):
            path = os.path.join(td, tag + ".npz")
            subprocess.run([sys.executable, "-c", code, path], check=True, env={**os.environ, **env}, timeout=600)
            res[tag] = dict(np.load(path))
    for k in res["fused"]:
        a, b = res["fused"][k], res["unfused"][k]
        if k.startswith("f"):
            # where the two forms name different bins, the fused form's own plane holds a value within 2 tol of its row
            # maximum at the other form's bin: a float32 tie
            pl = res["fused"]["p" + k[1:]]
            for r in np.nonzero(a != b)[0]:
                assert pl[r, b[r]] >= pl[r].max() - 2 * 2e-6, (k, r)
        else:
            assert np.max(np.abs(a - b)) <= 2e-6, k
