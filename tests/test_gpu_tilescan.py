"""GPU tests of the tiled scans: the float64 energy prefix behind every normalised output (k_prefix_tiles; the
reference's double-accumulated window energy, multiplySlices.cu:153,201 / filter.cu:324-339) and the ordered compaction
of findLocalMaxima (k_local_max_flags / k_local_max_write; peakfinding.cu:14-58), at the tile boundaries, with the
tile totals added up directly and scanned, on 16-byte-aligned and unaligned inputs, and run to run bit for bit."""

import ctypes as ct

import numpy as np
import pytest

from conftest import cn

pytestmark = pytest.mark.gpu

GROUP = 64  # (boundaries are probed at multiples of 64 tiles as well: one wave's worth of tile totals)
LM_TILE = 16384  # samples per tile of the local-maxima compaction (caf_rows.hip LM_TILE)


def prefix_tile(m):
    """samples per tile of the energy prefix for a record of m samples (caf_kernels.hip prefix_tile)"""
    return 2048


def _perdelay(d_cut, n, d_rx, m, start, num):
    from pydsproutines_amd import _lib
    from pydsproutines_amd.devarray import empty

    lib = _lib.load()
    qf2, idx = empty(num, np.float32), empty(num, np.int32)
    p = lambda a: ct.c_void_p(a.ptr)  # noqa: E731
    _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), m, int(start), 1, int(num), 0, p(qf2), p(idx), None, None, 0, None),
               "caf_xcorr_perdelay")
    _lib.check(lib.caf_stream_sync(None))
    return qf2.get(), idx.get()


@pytest.mark.parametrize("offset", [0, 1])
@pytest.mark.parametrize("m", [1500, 2047 + 64, 2048 * GROUP + 64, 2048 * GROUP * 3 + 1500, (1 << 20) + 5000, 8192 * GROUP * 3 + 1500,
                               8192 * 4097 + 1500])
def test_window_energy_across_tile_and_group_boundaries(m, offset):
    """qf2 = max_k |FFT(x[d:d+n] conj(c))|^2 / (E_c E_x[d]) with E_x from the prefix: windows that straddle every kind of
    boundary of the scan, on a record whose level changes by 60 dB along its length (a wrong tile offset is a wrong E_x)."""
    from pydsproutines_amd import asarray

    n = 64
    rng = np.random.default_rng(m + offset)
    level = 10.0 ** (-3.0 * np.arange(m + offset) / (m + offset))  # 0 ... -60 dB
    full = (cn(rng, m + offset) * level).astype(np.complex64)
    cut = cn(rng, n)
    d_full = asarray(full)
    d_rx = d_full[offset:]  # offset 1: an 8-byte-aligned record (the float2 path of the loader)
    rx = full[offset:]
    d_cut = asarray(cut.conj())
    TILE = prefix_tile(m)
    bounds = sorted({0, m - n - 80} | {b for b in (TILE, 2 * TILE, TILE * GROUP, 2 * TILE * GROUP, 3 * TILE * GROUP, 256 * TILE, 257 * TILE,
                                                  4096 * TILE) if b + 20 < m - n})
    ec = float(np.sum(np.abs(cut.astype(np.complex128)) ** 2))
    for b in bounds:
        start = max(0, b - 70)
        num = min(80, m - n + 1 - start)
        q, fi = _perdelay(d_cut, n, d_rx, m, start, num)
        q2, fi2 = _perdelay(d_cut, n, d_rx, m, start, num)
        np.testing.assert_array_equal(q, q2)  # run to run: the scan's float64 sums do not depend on timing
        np.testing.assert_array_equal(fi, fi2)
        for i in (0, num // 2, num - 1):
            w = rx[start + i : start + i + n].astype(np.complex128)
            spec = np.abs(np.fft.fft(w * cut.conj().astype(np.complex128))) ** 2
            want = spec.max() / (ec * np.sum(np.abs(w) ** 2))
            assert abs(q[i] - want) <= 2e-5 * max(1.0, want), (m, b, i)
            assert int(fi[i]) == int(np.argmax(spec))


@pytest.mark.parametrize("n", [1, 15, 16, 17, 4095, 4096, 4097, LM_TILE - 1, LM_TILE, LM_TILE + 1, GROUP * LM_TILE - 1, GROUP * LM_TILE,
                               GROUP * LM_TILE + 1, 3 * GROUP * LM_TILE + 5, (1 << 24) + 3])
def test_local_maxima_at_tile_and_group_boundaries(n):
    """Every other sample a maximum (the densest a trace can be: count = ~n/2, every lane writes), on aligned and
    unaligned traces, truncated and not; the count is the total found either way (peakfinding.cu's counter)."""
    from pydsproutines_amd import asarray
    from pydsproutines_amd.cupyExtensions import cupyFindLocalMaxima

    rng = np.random.default_rng(n)
    for offset in (0, 1, 3):
        full = np.zeros(n + offset, np.float32)
        v = full[offset:]
        v[::2] = 1.0 + rng.random(v[::2].size, dtype=np.float32)  # maxima at the even indices ...
        flat = rng.integers(0, max(1, n - 1), max(1, n // 50))
        v[flat] = v[np.minimum(flat + 1, n - 1)]  # ... except where a neighbour is equal (strict inequalities)
        y, l, r = v, np.concatenate(([0], v[:-1])), np.concatenate((v[1:], [0]))
        ref = np.flatnonzero((y > 0.5) & (y > l) & (y > r)).astype(np.int32)
        dv = asarray(full)[offset:]
        idx, cnt = cupyFindLocalMaxima(dv, 0.5, maxNumPeaks=max(1, ref.size))
        assert int(cnt.get()[0]) == ref.size
        np.testing.assert_array_equal(idx.get()[: ref.size], ref)
        if ref.size > 10:
            idx2, cnt2 = cupyFindLocalMaxima(dv, 0.5, maxNumPeaks=7)
            assert int(cnt2.get()[0]) == ref.size
            np.testing.assert_array_equal(idx2.get(), ref[:7])
    # nothing above the height: count 0, nothing written
    idx, cnt = cupyFindLocalMaxima(asarray(np.zeros(n, np.float32)), 0.5, maxNumPeaks=4)
    assert int(cnt.get()[0]) == 0


def test_local_maxima_with_scanned_tile_counts():
    """More than 4096 tiles: the counts of the preceding tiles come from a scan launch of their own (caf_rows.hip
    LM_DIRECT_TILES); sparse maxima, most tiles empty, the last one not."""
    from pydsproutines_amd import asarray
    from pydsproutines_amd.cupyExtensions import cupyFindLocalMaxima

    n = 4097 * LM_TILE + 3
    rng = np.random.default_rng(11)
    v = np.zeros(n, np.float32)
    pos = np.unique(np.concatenate((rng.integers(0, n, 5000), [0, LM_TILE - 1, LM_TILE, n - 1, n - 3, 4096 * LM_TILE, 4096 * LM_TILE - 1])))
    v[pos] = 1.0 + rng.random(pos.size, dtype=np.float32)
    l, r = np.concatenate(([0], v[:-1])), np.concatenate((v[1:], [0]))
    ref = np.flatnonzero((v > 0.5) & (v > l) & (v > r))
    idx, cnt = cupyFindLocalMaxima(asarray(v), 0.5, maxNumPeaks=ref.size)
    assert int(cnt.get()[0]) == ref.size
    np.testing.assert_array_equal(idx.get(), ref)


def test_scans_repeat_on_recycled_scratch():
    """The workspace comes from the pool uninitialised and is refilled before every launch: back-to-back calls of
    different sizes on the same stream."""
    from pydsproutines_amd import asarray
    from pydsproutines_amd.cupyExtensions import cupyFindLocalMaxima

    rng = np.random.default_rng(5)
    for n in (300000, 5000, 1 << 20, 17, 300000):
        v = np.abs(rng.standard_normal(n)).astype(np.float32)
        l, r = np.concatenate(([0], v[:-1])), np.concatenate((v[1:], [0]))
        ref = np.flatnonzero((v > 1.0) & (v > l) & (v > r))
        idx, cnt = cupyFindLocalMaxima(asarray(v), 1.0, maxNumPeaks=max(1, ref.size))
        assert int(cnt.get()[0]) == ref.size
        np.testing.assert_array_equal(idx.get()[: ref.size], ref)
