"""GPU parity tests of the hypothesis engine (caf_plan_* through the C-ABI) against the oracle
and the committed golden vectors.  Tolerance: CAF surface within 1e-4 of the surface
maximum; global (delay, frequency) peak exact; per-delay argmax exact wherever the oracle's top-2
margin exceeds the tolerance."""

import os

import numpy as np
import pytest

import oracle as O
from conftest import cn, qpsk

pytestmark = pytest.mark.gpu


def _surface_check(surf, ref, rel=1e-4):
    tol = rel * float(ref.max())
    err = float(np.max(np.abs(surf - ref)))
    assert err <= tol, "surface error %.3e > %.3e" % (err, tol)
    return tol


def _argmax_check(row_arg, row_max, ref, tol):
    """Exact argmax where the oracle's top-2 margin is above tolerance; value always within tol."""
    top2 = np.sort(ref, axis=1)[:, -2:] if ref.shape[1] > 1 else np.hstack((np.full((ref.shape[0], 1), -1.0), ref))
    clear = (top2[:, 1] - top2[:, 0]) > 2 * tol
    np.testing.assert_array_equal(row_arg[clear], np.argmax(ref, axis=1)[clear])
    assert np.max(np.abs(row_max - ref.max(axis=1))) <= tol
    return int(clear.sum())


def test_c2_mini_golden(golden):
    from pydsproutines_amd import CAFPlan, asarray

    g = golden("c2_mini")
    t, rx, bins, sh = g["template"], g["rx"], g["bins"], g["shifts"]
    plan = CAFPlan(t, max_rx_len=rx.size, bins=bins, grid=t.size, log2_block=10)
    res = plan.run(asarray(rx), surface=True)
    surf = res.surface.get()[0]
    assert surf.shape == (rx.size - t.size + 1, bins.size)
    tol = _surface_check(surf[sh], g["caf"])
    n_clear = _argmax_check(res.row_arg.get()[0][sh], res.row_max.get()[0][sh], g["caf"], tol)
    assert n_clear > sh.size // 2
    assert int(res.peak_delay.get()[0]) == int(g["d0"][0])
    assert int(bins[res.peak_freq.get()[0]]) == int(g["k0"][0])
    assert abs(float(res.peak_val.get()[0]) - g["caf"].max()) <= tol
    # row results are exactly the max / first argmax of the surface the GPU itself wrote
    np.testing.assert_array_equal(res.row_max.get()[0], surf.max(axis=1))
    np.testing.assert_array_equal(res.row_arg.get()[0], np.argmax(surf, axis=1))


@pytest.mark.parametrize("log2_block,nb", [(9, 1), (10, 3), (12, 2), (0, 0)])
@pytest.mark.parametrize("F", [1, 5, 32, 200])
def test_engine_vs_oracle_bins(log2_block, nb, F):
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(100 + F)
    n, m = 128, 5000 + 37 * F
    t = qpsk(rng, n)
    rx = cn(rng, m)
    d0, k0 = 2345, int(F // 3) - F // 2
    rx[d0 : d0 + n] += (1.5 * t * np.exp(2j * np.pi * k0 * np.arange(n) / n)).astype(np.complex64)
    bins = np.arange(F) - F // 2
    plan = CAFPlan(t, max_rx_len=m, bins=bins, grid=n, log2_block=log2_block, blocks_per_batch=nb)
    res = plan.run(asarray(rx), surface=True)
    ref = O.caf_bins(t, rx, bins)
    surf = res.surface.get()[0]
    tol = _surface_check(surf, ref)
    _argmax_check(res.row_arg.get()[0], res.row_max.get()[0], ref, tol)
    r, c = np.unravel_index(np.argmax(ref), ref.shape)
    assert (int(res.peak_delay.get()[0]), int(res.peak_freq.get()[0])) == (r, c) == (d0, k0 + F // 2)


def test_shift_range_and_peak_only():
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(5)
    n, m = 200, 9000  # non power-of-two template, grid = 256-point DFT bins
    t = cn(rng, n)
    rx = cn(rng, m)
    rx[4000 : 4000 + n] += 2 * t
    bins = np.array([0, 3, -3, 17, 100, -128], dtype=np.int32)  # unsorted, sparse, odd shifts
    plan = CAFPlan(t, max_rx_len=m, bins=bins, grid=256, log2_block=11)
    d_rx = asarray(rx)
    res = plan.run(d_rx, shift_start=3900, num_shifts=333, surface=False, rows=True, peak=True)
    assert res.surface is None
    # oracle: 256-point DFT of the (zero-padded) 200-sample product
    def ref_grid(sh):
        win = np.lib.stride_tricks.sliding_window_view(rx, n)[sh].astype(np.complex128)
        spec = np.fft.fft(win * t.conj().astype(np.complex128), 256, axis=1)[:, np.mod(bins, 256)]
        e = np.sum(np.abs(win) ** 2, axis=1)
        return np.abs(spec) ** 2 / e[:, None] / np.sum(np.abs(t.astype(np.complex128)) ** 2)

    ref = ref_grid(np.arange(3900, 3900 + 333))
    tol = 1e-4 * ref.max()
    _argmax_check(res.row_arg.get()[0], res.row_max.get()[0], ref, tol)
    assert int(res.peak_delay.get()[0]) == 4000 and int(res.peak_freq.get()[0]) == 0
    # re-running with a shorter range after a longer one must not see stale partial peaks
    res2 = plan.run(d_rx, shift_start=100, num_shifts=50)
    ref2 = ref_grid(np.arange(100, 150))
    assert abs(float(res2.peak_val.get()[0]) - ref2.max()) <= 1e-4 * ref2.max()
    r2, c2 = np.unravel_index(np.argmax(ref2), ref2.shape)
    assert (int(res2.peak_delay.get()[0]), int(res2.peak_freq.get()[0])) == (100 + r2, c2)


@pytest.mark.parametrize("T,F", [(2, 201), (1, 65), (3, 64), (2, 130)])
def test_no_surface_mode_any_frequency_count(T, F):
    """Per-delay traces and peaks without a surface (running maxima in the FFT role, hypothesis groups formed per
    template, e.g. F = 201 -> 51/51/51/48) are bit-identical to the ones of the surface run, for frequency counts
    that are and are not multiples of the 64-hypothesis group; CZT-like explicit frequency grid."""
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(7 * T + F)
    n, m = 512, 300_000
    tm = np.stack([qpsk(rng, n) for _ in range(T)])
    rx = cn(rng, m)
    freqs = (np.arange(F) - F // 2) * (0.25 / n)
    for i in range(T):
        rx[50_000 * (i + 1) : 50_000 * (i + 1) + n] += (2 * tm[i] * np.exp(2j * np.pi * freqs[(7 * i + 3) % F] * np.arange(n))).astype(np.complex64)
    plan = CAFPlan(tm, max_rx_len=m, freqs_norm=freqs, engine="persistent")
    d_rx = asarray(rx)
    a = plan.run(d_rx, surface=True)
    want = (a.row_max.get(), a.row_arg.get(), a.peak_val.get(), a.peak_delay.get(), a.peak_freq.get())
    np.testing.assert_array_equal(want[0], a.surface.get().max(axis=2))
    b = plan.run(d_rx, surface=False, rows=True, peak=True)
    for x, y in zip(want, (b.row_max.get(), b.row_arg.get(), b.peak_val.get(), b.peak_delay.get(), b.peak_freq.get())):
        np.testing.assert_array_equal(x, y)
    for i in range(T):
        assert int(want[3][i]) == 50_000 * (i + 1) and int(want[4][i]) == (7 * i + 3) % F
    c = plan.run(d_rx, shift_start=1234, num_shifts=100_001, surface=False, rows=False, peak=True)
    j = np.argmax(want[0][:, 1234 : 1234 + 100_001], axis=1)
    np.testing.assert_array_equal(c.peak_delay.get(), 1234 + j)
    plan.close()


def test_freqs_norm_table_mode_vs_groupxcorr(golden):
    """Arbitrary (off-grid) frequencies + composite template == GroupXcorr (xcorrRoutines.py:852-954)."""
    from pydsproutines_amd import CAFPlan, asarray

    g = golden("groupxcorr_small")
    fs = float(g["fs"][0])
    y, starts, lengths, rx, freqs, sh = g["y"], g["starts"], g["lengths"], g["rx"], g["freqs"], g["shifts"]
    rel = starts - starts[0]
    span = int(rel[-1] + lengths[-1])
    tm = np.zeros(span, np.complex64)
    for s, r, l in zip(starts, rel, lengths):
        tm[r : r + l] = y[s : s + l]
    plan = CAFPlan(tm, max_rx_len=rx.size, freqs_norm=freqs / fs, group_starts=rel, group_lens=lengths, log2_block=11)
    res = plan.run(asarray(rx), shift_start=int(sh[0]), num_shifts=sh.size, surface=True)
    ref = O.GroupXcorr(y, starts, lengths, freqs, fs).caf(rx, sh)
    surf = res.surface.get()[0]
    tol = _surface_check(surf, ref)
    np.testing.assert_allclose(res.row_max.get()[0], g["xc"], atol=tol)
    clear = np.sort(ref, axis=1)[:, -1] - np.sort(ref, axis=1)[:, -2] > 2 * tol
    np.testing.assert_array_equal(freqs[res.row_arg.get()[0]][clear], g["freqpeaks"][clear])
    assert int(res.peak_delay.get()[0]) == 777 and freqs[int(res.peak_freq.get()[0])] == 6.0


def test_multi_template_and_host_call():
    from pydsproutines_amd import CAFPlan

    rng = np.random.default_rng(9)
    n, m, T = 64, 3000, 5
    tm = np.stack([qpsk(rng, n) * (1 + i) for i in range(T)])
    rx = (0.7 * cn(rng, m)).astype(np.complex64)
    delays = [100, 700, 1500, 2200, 2900]
    for i, d in enumerate(delays):
        rx[d : d + n] += tm[i] / (1 + i)
    plan = CAFPlan(tm, max_rx_len=m, bins=[0, 1, -1], grid=n, log2_block=9, blocks_per_batch=2)
    out = plan.run_host(rx, surface=True)
    for i in range(T):
        ref = O.caf_bins(tm[i], rx, [0, 1, -1])
        tol = _surface_check(out["surface"][i], ref)
        assert int(out["peak_delay"][i]) == delays[i] and int(out["peak_freq"][i]) == 0
        assert abs(out["peak_val"][i] - ref.max()) <= tol


def test_c1_full(golden):
    """Config C1: 1024-sample template vs 65536-sample rx, no frequency search (F=1, bin 0)."""
    from pydsproutines_amd import CAFPlan, asarray

    g = golden("c1_fastxcorr")
    rx, d0 = g["rx"], int(g["d0"][0])
    plan = CAFPlan(rx[d0 : d0 + 1024].copy(), max_rx_len=rx.size, bins=[0], grid=1024)
    res = plan.run(asarray(rx))
    q = res.row_max.get()[0]
    assert q.shape == (64513,)
    assert np.max(np.abs(q - g["qf2"])) <= 1e-6 + 1e-4 * 1.0 * 0.01  # 1e-6 abs (same arithmetic class)
    assert int(res.peak_delay.get()[0]) == d0 and abs(float(res.peak_val.get()[0]) - 1.0) <= 1e-5


def test_argument_validation():
    from pydsproutines_amd import CAFPlan, asarray

    t = np.ones(64, np.complex64)
    with pytest.raises(ValueError):
        CAFPlan(t, max_rx_len=32, bins=[0], grid=64)  # rx shorter than template
    with pytest.raises(ValueError):
        CAFPlan(t, max_rx_len=1000, bins=[0, 1], grid=48)  # bin 1 of a 48-point grid is no whole element of the block
    with pytest.raises(ValueError):
        CAFPlan(t, max_rx_len=1000, bins=[0], grid=64, log2_block=5)  # block < 2N
    with pytest.raises(ValueError):
        CAFPlan(t, max_rx_len=1000)  # neither bins nor freqs
    plan = CAFPlan(t, max_rx_len=1000, bins=[0], grid=64)
    rx = asarray(np.ones(500, np.complex64))
    with pytest.raises(ValueError):
        plan.run(rx, shift_start=400, num_shifts=100)  # runs past the end
    with pytest.raises(ValueError):
        plan.run(asarray(np.ones(2000, np.complex64)))  # longer than max_rx_len
    with pytest.raises(TypeError):
        plan.run(np.ones(500, np.complex64))  # host array where a device array is required
    # 65536 hypotheses: their |y|^2 tiles would need more than 32-bit byte offsets inside a block -- refused with a
    # message for surface / tile launches, still served without a surface (running maxima, no tiles)
    many = CAFPlan(t, max_rx_len=400, bins=np.arange(65536) % 64, grid=64)
    small = asarray((np.arange(400) % 7).astype(np.complex64))
    with pytest.raises(ValueError, match="too many hypotheses"):
        many.run(small, surface=True)
    res = many.run(small, surface=False, rows=True, peak=True)
    assert res.row_max.shape == (1, 337) and int(res.row_arg.get().max()) < 65536


@pytest.mark.parametrize("engine", ["fused", "persistent", "rocfft"])
def test_engines_agree_with_oracle(engine, golden):
    """All inverse-transform engines (hand-written LDS FFT kernel as two launches / as one work-queue
    launch / rocFFT) against the oracle: on-grid shift mode, explicit-frequency table mode with groups,
    multi-template, ragged tail."""
    from pydsproutines_amd import CAFPlan, asarray

    g = golden("c2_mini")
    t, rx, bins, sh = g["template"], g["rx"], g["bins"], g["shifts"]
    plan = CAFPlan(t, max_rx_len=rx.size, bins=bins, grid=t.size, engine=engine)
    assert plan.engine_used == engine
    assert plan.block == (16384 if engine != "rocfft" else plan.block)
    res = plan.run(asarray(rx), surface=True)
    surf = res.surface.get()[0]
    tol = _surface_check(surf[sh], g["caf"])
    _argmax_check(res.row_arg.get()[0][sh], res.row_max.get()[0][sh], g["caf"], tol)
    assert (int(res.peak_delay.get()[0]), int(bins[res.peak_freq.get()[0]])) == (int(g["d0"][0]), int(g["k0"][0]))
    np.testing.assert_array_equal(res.row_max.get()[0], surf.max(axis=1))
    np.testing.assert_array_equal(res.row_arg.get()[0], np.argmax(surf, axis=1))

    # several blocks of 16384 with a ragged tail, 3 templates x 70 bins (2 chunks of hypotheses)
    rng = np.random.default_rng(77)
    n, m, T = 512, 40000, 3
    tm = np.stack([qpsk(rng, n) for _ in range(T)])
    rx2 = cn(rng, m)
    spots = [(1000, 3), (17000, -20), (39000, 34)]
    for i, (d0, k0) in enumerate(spots):
        rx2[d0 : d0 + n] += (2 * tm[i] * np.exp(2j * np.pi * k0 * np.arange(n) / n)).astype(np.complex64)
    b2 = np.arange(-35, 35)
    plan2 = CAFPlan(tm, max_rx_len=m, bins=b2, grid=n, engine=engine, blocks_per_batch=2)
    r2 = plan2.run(asarray(rx2), surface=True)
    rows = np.concatenate((np.arange(0, 200), np.arange(15800, 17100), np.arange(m - n + 1 - 300, m - n + 1)))
    for i, (d0, k0) in enumerate(spots):
        ref = O.caf_bins(tm[i], rx2, b2, rows)
        s_i = r2.surface.get()[i]
        tol = _surface_check(s_i[rows], ref)
        _argmax_check(r2.row_arg.get()[i][rows], r2.row_max.get()[i][rows], ref, tol)
        assert (int(r2.peak_delay.get()[i]), int(b2[r2.peak_freq.get()[i]])) == (d0, k0)

    # table mode + groups
    gg = golden("groupxcorr_small")
    fs = float(gg["fs"][0])
    y, starts, lengths, rxg, freqs, shg = gg["y"], gg["starts"], gg["lengths"], gg["rx"], gg["freqs"], gg["shifts"]
    rel = starts - starts[0]
    comp = np.zeros(int(rel[-1] + lengths[-1]), np.complex64)
    for s, r, l in zip(starts, rel, lengths):
        comp[r : r + l] = y[s : s + l]
    plan3 = CAFPlan(comp, max_rx_len=rxg.size, freqs_norm=freqs / fs, group_starts=rel, group_lens=lengths, engine=engine)
    r3 = plan3.run(asarray(rxg), shift_start=int(shg[0]), num_shifts=shg.size, surface=True)
    ref3 = O.GroupXcorr(y, starts, lengths, freqs, fs).caf(rxg, shg)
    _surface_check(r3.surface.get()[0], ref3)
    assert int(r3.peak_delay.get()[0]) == 777 and freqs[int(r3.peak_freq.get()[0])] == 6.0
    if engine != "rocfft":
        with pytest.raises(ValueError):
            plan3.run(asarray(rxg), cqf=True)  # complex QF is a rocFFT-engine output
        with pytest.raises(ValueError):  # (beyond the eight partitions of 32768 samples; the two-launch engine ends at 8192)
            CAFPlan(np.ones(262145, np.complex64), max_rx_len=400000, bins=[0], grid=16384, engine=engine)
        with pytest.raises(ValueError):
            CAFPlan(np.ones(33000, np.complex64), max_rx_len=80000, bins=[0], grid=16384, engine="fused")

    # 2 templates x 128 bins (whole 128-hypothesis chunks: the full-tile path of the transposers), 3 blocks,
    # run twice on different data through the same plan: the second result must not see the first one's tiles
    F4 = 128
    b4 = np.arange(-F4 // 2, F4 // 2)
    tm4 = np.stack([qpsk(rng, n) for _ in range(2)])
    plan4 = CAFPlan(tm4, max_rx_len=m, bins=b4, grid=n, engine=engine)
    for trial, spots4 in enumerate([[(2000, -60), (30000, 63)], [(25000, 5), (123, -1)]]):
        rx4 = cn(rng, m)
        for i, (d0, k0) in enumerate(spots4):
            rx4[d0 : d0 + n] += (2 * tm4[i] * np.exp(2j * np.pi * k0 * np.arange(n) / n)).astype(np.complex64)
        r4 = plan4.run(asarray(rx4), surface=True)
        rows4 = np.concatenate((np.arange(0, 300), np.arange(15700, 16200), np.arange(24900, 25100), np.arange(m - n + 1 - 200, m - n + 1)))
        for i, (d0, k0) in enumerate(spots4):
            ref = O.caf_bins(tm4[i], rx4, b4, rows4)
            s_i = r4.surface.get()[i]
            tol = _surface_check(s_i[rows4], ref)
            _argmax_check(r4.row_arg.get()[i][rows4], r4.row_max.get()[i][rows4], ref, tol)
            np.testing.assert_array_equal(r4.row_max.get()[i], s_i.max(axis=1))
            np.testing.assert_array_equal(r4.row_arg.get()[i], np.argmax(s_i, axis=1))
            assert (int(r4.peak_delay.get()[i]), int(b4[r4.peak_freq.get()[i]])) == (d0, k0)
        # peak-only call (no surface, no rows) gives the same peaks
        r5 = plan4.run(asarray(rx4), surface=False, rows=False, peak=True)
        np.testing.assert_array_equal(r5.peak_delay.get(), r4.peak_delay.get())
        np.testing.assert_array_equal(r5.peak_val.get(), r4.peak_val.get())

    # no frequency scan (one bin), 37 templates (more than one 32-template step, ragged): the C3 / template-bank
    # shape, which the persistent engine streams without a transpose
    T6 = 37
    tm6 = np.stack([qpsk(rng, n) for _ in range(T6)])
    rx6 = cn(rng, m)
    d6 = 500 + 1000 * np.arange(T6) + rng.integers(0, 64, T6)
    for i in range(T6):
        rx6[d6[i] : d6[i] + n] += 2 * tm6[i]
    plan6 = CAFPlan(tm6, max_rx_len=m, bins=[0], grid=n, engine=engine)
    r6 = plan6.run(asarray(rx6), surface=True)
    np.testing.assert_array_equal(r6.peak_delay.get(), d6)
    assert np.all(r6.row_arg.get() == 0) and np.all(r6.peak_freq.get() == 0)
    np.testing.assert_array_equal(r6.surface.get()[:, :, 0], r6.row_max.get())
    rm6 = r6.row_max.get()
    for i in (0, 17, 36):
        ref = O.fastXcorr(tm6[i], rx6)
        assert np.max(np.abs(rm6[i] - ref)) <= 1e-4 * ref.max()
        assert float(r6.peak_val.get()[i]) == rm6[i].max()
    r7 = plan6.run(asarray(rx6), shift_start=700, num_shifts=20001, surface=False, rows=True, peak=True)
    np.testing.assert_allclose(r7.row_max.get(), rm6[:, 700:20701], atol=2e-6)
    assert np.all(r7.peak_delay.get()[1:20] == d6[1:20])


@pytest.mark.parametrize("n", [4095, 8191, 8192, 8193, 12000, 16384, 16385, 24000, 32768, 32769, 49152, 65536, 65537, 100000,
                               131072, 131073, 200000, 262144, 262145])
def test_template_lengths_around_the_fused_limits(n):
    """The LDS-resident engines: 16384-point blocks for templates up to 8192 samples (persistent and two-launch fused,
    bit-identical), 32768-point blocks as two chained 16384-point transforms up to 16384 samples, 65536-point blocks in the
    folded form (two chained transforms per output residue) up to 32768 samples (persistent only), the same blocks with the
    template cut into 2 .. 8 partitions of 32768 samples up to 262144; longer ones go to the rocfft engine automatically and
    are refused by an explicit fused / persistent request."""
    from pydsproutines_amd import CAFPlan, asarray
    from test_gpu_engine_fuzz import _oracle_rows

    rng = np.random.default_rng(n)
    m = 40_000 if n <= 8192 else 90_000 if n <= 16384 else 150_000 if n <= 32768 else n + 110_000
    t = qpsk(rng, n)
    rx = cn(rng, m)
    d0, bins = 12_345, np.arange(-4, 4)
    grid = min(16384, 1 << int(np.ceil(np.log2(n))))
    rx[d0 : d0 + n] += (t * np.exp(2j * np.pi * 3 * np.arange(n) / grid)).astype(np.complex64)
    d_rx = asarray(rx)
    rows = np.array([0, 1, d0 - 1, d0, d0 + 1, 16383, 16384, 16385, 24575, 24576, 32767, 32768, 49151, 49152, 65535, 65536,
                     98303, 98304, m - n])
    rows = rows[rows <= m - n]
    ref = _oracle_rows(t, rx, bins / grid, rows)
    engines = ("persistent", "fused", "rocfft") if n <= 8192 else ("persistent", "rocfft") if n <= 262144 else ("rocfft",)
    surf = {}
    for engine in engines:
        plan = CAFPlan(t, max_rx_len=m, bins=bins, grid=grid, engine=engine)
        res = plan.run(d_rx, surface=True)
        surf[engine] = res.surface.get()[0]
        np.testing.assert_allclose(surf[engine][rows], ref, atol=1e-4 * ref.max())
        assert (int(res.peak_delay.get()[0]), int(bins[res.peak_freq.get()[0]])) == (d0, 3)
        # row results == the surface it wrote; peak-only run agrees
        np.testing.assert_array_equal(res.row_max.get()[0], surf[engine].max(axis=1))
        np.testing.assert_array_equal(res.row_arg.get()[0], np.argmax(surf[engine], axis=1))
        r2 = plan.run(d_rx, surface=False)
        np.testing.assert_array_equal(r2.row_max.get()[0], surf[engine].max(axis=1))
        plan.close()
    if n <= 8192:
        np.testing.assert_array_equal(surf["persistent"], surf["fused"])
        auto = CAFPlan(t, max_rx_len=m, bins=bins, grid=grid)
        assert auto.engine_used == "persistent" and auto.block == 16384
        auto.close()
    elif n <= 16384:
        assert np.max(np.abs(surf["persistent"] - surf["rocfft"])) <= 2e-6 * max(1.0, surf["rocfft"].max())  # two engines
        auto = CAFPlan(t, max_rx_len=m, bins=bins, grid=grid)
        valid = 32768 - n + 1  # a step a few delays over a multiple of 64 is rounded down to whole 64-delay tiles
        assert auto.engine_used == "persistent" and auto.block == 32768 and auto.step in (valid, valid - valid % 64)
        auto.close()
        with pytest.raises(ValueError):
            CAFPlan(t, max_rx_len=m, bins=bins, grid=grid, engine="fused")
    elif n <= 262144:
        assert np.max(np.abs(surf["persistent"] - surf["rocfft"])) <= 2e-6 * max(1.0, surf["rocfft"].max())  # two engines
        auto = CAFPlan(t, max_rx_len=m, bins=bins, grid=grid)
        assert auto.engine_used == "persistent" and auto.block == 65536 and auto.step == 32768
        auto.close()
        with pytest.raises(ValueError):
            CAFPlan(t, max_rx_len=m, bins=bins, grid=grid, engine="fused")
    else:
        auto = CAFPlan(t, max_rx_len=m, bins=bins, grid=grid)
        assert auto.engine_used == "rocfft"
        auto.close()
        for engine in ("persistent", "fused"):
            with pytest.raises(ValueError):
                CAFPlan(t, max_rx_len=m, bins=bins, grid=grid, engine=engine)


@pytest.mark.parametrize("n", [20000, 40000, 70000])
def test_chained_roles_one_plan_many_rx_lengths(n):
    """A plan sized for the longest rx it will see, run on shorter ones and on delay sub-ranges in any order (the partitioned role
    reads the spectra of the blocks BEYOND a call's last one: they must be this call's, zero-padded past the end of its rx, not
    what a longer call left behind): every call equals a plan created for exactly that rx."""
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(n)
    t = qpsk(rng, n)
    bins = np.arange(-2, 3)
    big = n + 200_000
    rx = cn(rng, big)
    plan = CAFPlan(t, max_rx_len=big, bins=bins, grid=16384)
    assert plan.engine_used == "persistent"
    for m, lo, cnt in ((big, 0, None), (n + 70_000, 0, None), (n + 131_072, 5, 100_000), (n + 1, 0, None), (big, 150_000, 50_001),
                       (n + 32_768, 0, None)):
        d = asarray(rx[:m].copy())
        got = plan.run(d, shift_start=lo, num_shifts=cnt, surface=True)
        ref_plan = CAFPlan(t, max_rx_len=m, bins=bins, grid=16384)
        ref = ref_plan.run(d, shift_start=lo, num_shifts=cnt, surface=True)
        assert np.array_equal(got.surface.get(), ref.surface.get()), (m, lo, cnt)
        assert np.array_equal(got.row_arg.get(), ref.row_arg.get()) and np.array_equal(got.peak_delay.get(), ref.peak_delay.get())
        ref_plan.close()
    plan.close()


def test_long_template_explicit_frequencies_and_many_hypotheses():
    """32768-point blocks with an explicit frequency table (off-grid hypotheses, table mode) and more hypotheses than
    one work item takes, several templates: against the rocfft engine and the oracle."""
    from pydsproutines_amd import CAFPlan, asarray
    from test_gpu_engine_fuzz import _oracle_rows

    rng = np.random.default_rng(5)
    n, m = 10_000, 120_000
    tm = np.stack([qpsk(rng, n) for _ in range(2)])
    rx = cn(rng, m)
    freqs = np.linspace(-3.3e-4, 3.1e-4, 150)
    truth = [(20_000, 17), (77_777, 140)]
    for i, (d, f) in enumerate(truth):
        rx[d : d + n] += (tm[i] * np.exp(2j * np.pi * freqs[f] * np.arange(n))).astype(np.complex64)
    d_rx = asarray(rx)
    p = CAFPlan(tm, max_rx_len=m, freqs_norm=freqs)
    assert p.engine_used == "persistent" and p.block == 32768
    r = p.run(d_rx, surface=True)
    q = CAFPlan(tm, max_rx_len=m, freqs_norm=freqs, engine="rocfft").run(d_rx, surface=True)
    a, b = r.surface.get(), q.surface.get()
    assert np.max(np.abs(a - b)) <= 3e-6 * max(1.0, b.max())
    for i, (d, f) in enumerate(truth):
        assert (int(r.peak_delay.get()[i]), int(r.peak_freq.get()[i])) == (d, f)
        rows = np.array([d - 1, d, d + 1, 50_000])
        ref = _oracle_rows(tm[i], rx, freqs, rows)
        np.testing.assert_allclose(a[i][rows], ref, atol=1e-4 * ref.max())


def test_65536_point_role_explicit_frequencies_and_several_templates():
    """65536-point blocks (templates of 16385 .. 32768 samples) with an explicit frequency table (table mode: one row of
    parity pairs per hypothesis), more hypotheses than one work item takes and an odd number of them, two templates, a sub-range: against
    the rocfft engine and the oracle."""
    from pydsproutines_amd import CAFPlan, asarray
    from test_gpu_engine_fuzz import _oracle_rows

    rng = np.random.default_rng(6)
    n, m = 20_000, 200_000
    tm = np.stack([qpsk(rng, n) for _ in range(2)])
    rx = cn(rng, m)
    freqs = np.linspace(-1.7e-4, 1.6e-4, 75)
    truth = [(30_000, 11), (140_017, 70)]
    for i, (d, f) in enumerate(truth):
        rx[d : d + n] += (tm[i] * np.exp(2j * np.pi * freqs[f] * np.arange(n))).astype(np.complex64)
    d_rx = asarray(rx)
    p = CAFPlan(tm, max_rx_len=m, freqs_norm=freqs)
    assert p.engine_used == "persistent" and p.block == 65536
    r = p.run(d_rx, surface=True)
    q = CAFPlan(tm, max_rx_len=m, freqs_norm=freqs, engine="rocfft").run(d_rx, surface=True)
    a, b = r.surface.get(), q.surface.get()
    assert np.max(np.abs(a - b)) <= 3e-6 * max(1.0, b.max())
    for i, (d, f) in enumerate(truth):
        assert (int(r.peak_delay.get()[i]), int(r.peak_freq.get()[i])) == (d, f)
        rows = np.array([d - 1, d, d + 1, 32767, 32768, 100_000])
        ref = _oracle_rows(tm[i], rx, freqs, rows)
        np.testing.assert_allclose(a[i][rows], ref, atol=1e-4 * ref.max())
    sub = p.run(d_rx, shift_start=25_000, num_shifts=40_001, surface=False)
    np.testing.assert_allclose(sub.row_max.get(), r.row_max.get()[:, 25_000:65_001], atol=2e-6)
    assert int(sub.peak_delay.get()[0]) == 30_000
    p.close()


def test_no_surface_items_of_up_to_256_hypotheses():
    """A launch without |y|^2 tiles that has work items to spare (>= 40 per CU) takes groups of up to 256 hypotheses instead of
    64 (the running maxima pack the index of a hypothesis within its group into 8 bits).  The same job with the groups pinned
    to 64 (CAF_HYP_PER_WG) must give the same per-delay maxima bit for bit, the same arguments (float32 ties between two
    hypotheses of one group aside) and the same peaks -- with planted peaks at hypotheses 0, 124, 125 and 249 of a template,
    i.e. at both ends of the two groups of 125 the rule forms for 250 bins (from five of 50)."""
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(256)
    n, T, F = 4096, 32, 250
    m = n + 176 * 12288
    bins = np.arange(-125, 125)
    tm = np.stack([qpsk(rng, n) for _ in range(T)])
    rx = cn(rng, m)
    truth = {0: (1_000, 0), 5: (500_000, 124), 17: (1_200_345, 125), 31: (m - n, 249)}
    for i, (d, j) in truth.items():
        rx[d : d + n] += (tm[i] * np.exp(2j * np.pi * bins[j] / n * np.arange(n))).astype(np.complex64)
    d_rx = asarray(rx)
    out = {}
    for tag, env in (("auto", None), ("pinned", "64")):
        if env is None:
            os.environ.pop("CAF_HYP_PER_WG", None)
        else:
            os.environ["CAF_HYP_PER_WG"] = env
        try:
            p = CAFPlan(tm, max_rx_len=m, bins=bins, grid=n)
            assert p.engine_used == "persistent"
            r = p.run(d_rx, surface=False, rows=True, peak=True)
            out[tag] = (r.row_max.get(), r.row_arg.get(), r.peak_val.get(), r.peak_delay.get(), r.peak_freq.get())
            p.close()
        finally:
            os.environ.pop("CAF_HYP_PER_WG", None)
    a, b = out["auto"], out["pinned"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert np.mean(a[1] == b[1]) > 0.9999 and np.array_equal(a[4], b[4])
    for i, (d, j) in truth.items():
        assert (int(a[3][i]), int(a[4][i])) == (d, j) and int(a[1][i][d]) == j


def test_65536_point_role_tiles_of_every_second_delay():
    """The folded 65536-point role leaves tiles of every second delay (tile 256 r + u = delays 2 (64 u + j) + r), so each tile
    role runs with a delay stride: no frequency scan (three templates, one bin: the streaming tile role), per-delay maxima
    without the surface, and the full surface -- with a ragged last block whose valid count is odd, sub-ranges that start at
    odd delays and end inside a tile, every template's row checked against the oracle at block / tile / parity boundaries."""
    from pydsproutines_amd import CAFPlan, asarray
    from test_gpu_engine_fuzz import _oracle_rows

    rng = np.random.default_rng(65)
    n, m = 17_001, 17_001 + 2 * 32768 + 4_320  # 69857 delays: two whole blocks + 4321 (an odd count) in the third
    S = m - n + 1
    tm = np.stack([qpsk(rng, n) for _ in range(3)])
    rx = cn(rng, m)
    truth = [(1, 0), (32_769, -2), (S - 1, 3)]  # the first odd row, the second block's first odd row, the very last row
    bins = np.arange(-2, 4)
    for i, (d, b) in enumerate(truth):
        rx[d : d + n] += (tm[i] * np.exp(2j * np.pi * b / 16384 * np.arange(n))).astype(np.complex64)
    d_rx = asarray(rx)
    rows = np.unique(np.concatenate((np.arange(0, 4), [126, 127, 128, 129, 32766, 32767, 32768, 32769, 65535, 65536, 65537],
                                     np.arange(S - 4, S), rng.integers(0, S, 12))))
    # (a) no frequency scan: the hypothesis axis of a tile is the template axis
    p1 = CAFPlan(tm, max_rx_len=m, bins=[0], grid=16384)
    assert p1.engine_used == "persistent" and p1.block == 65536 and p1.step == 32768
    r1 = p1.run(d_rx, surface=True)
    a1, rm1 = r1.surface.get(), r1.row_max.get()
    assert a1.shape == (3, S, 1) and np.array_equal(a1[:, :, 0], rm1) and not np.any(r1.row_arg.get())
    for i in range(3):
        ref = _oracle_rows(tm[i], rx, np.array([0.0]), rows)[:, 0]
        np.testing.assert_allclose(rm1[i][rows], ref, atol=1e-4 * max(ref.max(), 1e-3))
    assert int(r1.peak_delay.get()[0]) == truth[0][0]
    for s0, ns in ((1, S - 1), (32_767, 3), (4_001, 40_002), (S - 65, 65)):
        sub = p1.run(d_rx, shift_start=s0, num_shifts=ns, surface=False)
        np.testing.assert_allclose(sub.row_max.get(), rm1[:, s0 : s0 + ns], atol=2e-6)
        assert int(sub.peak_delay.get()[0]) == s0 + int(np.argmax(rm1[0, s0 : s0 + ns]))
    p1.close()
    # (b) six bins: surface, and the same rows / arguments / peaks without it
    p6 = CAFPlan(tm, max_rx_len=m, bins=bins, grid=16384)
    r6 = p6.run(d_rx, surface=True)
    a6 = r6.surface.get()
    for i, (d, b) in enumerate(truth):
        ref = _oracle_rows(tm[i], rx, bins / 16384.0, rows)
        np.testing.assert_allclose(a6[i][rows], ref, atol=1e-4 * ref.max())
        assert (int(r6.peak_delay.get()[i]), int(bins[r6.peak_freq.get()[i]])) == (d, b)
    assert np.array_equal(r6.row_max.get(), a6.max(axis=2)) and np.array_equal(r6.row_arg.get(), a6.argmax(axis=2))
    for s0, ns in ((0, S), (32_769, 32_768), (65_535, S - 65_535)):
        q = p6.run(d_rx, shift_start=s0, num_shifts=ns, surface=False)
        if s0 == 0:  # the same blocks: bit for bit; a shifted range cuts rx into other blocks (other roundings)
            assert np.array_equal(q.row_max.get(), r6.row_max.get()[:, s0 : s0 + ns])
            assert np.array_equal(q.row_arg.get(), r6.row_arg.get()[:, s0 : s0 + ns])
        else:
            np.testing.assert_allclose(q.row_max.get(), r6.row_max.get()[:, s0 : s0 + ns], atol=2e-6)
            assert np.mean(q.row_arg.get() == r6.row_arg.get()[:, s0 : s0 + ns]) > 0.999
        for i in range(3):
            want = s0 + int(np.argmax(r6.row_max.get()[i, s0 : s0 + ns]))
            assert int(q.peak_delay.get()[i]) == want
    p6.close()


def test_direct_engine_for_templates_with_a_handful_of_samples():
    """Composite templates whose groups cover fewer than 64 samples go to the direct (time-domain) engine: the
    overlap-save engines' float32 error follows the energy of a whole 16384-sample block, which for a normalisation
    over 16 samples is 3-6e-5; the definition evaluated product by product is at 1e-6.  Checked: AUTO picks it, its
    surface against the oracle's GroupXcorr plane at a tight tolerance, agreement with the persistent engine at the
    wide one, rows / peak consistent with its own surface, the no-surface call, explicit selection and refusal."""
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(2024)
    fs = 1000.0
    starts = np.array([0, 700, 2991])
    lengths = np.array([5, 2, 9])
    span = int(starts[-1] + lengths[-1])
    T, m = 2, 30000
    freqs = np.linspace(-40.0, 40.0, 37)
    ys = [cn(rng, span) for _ in range(T)]
    rx = cn(rng, m)
    comps = []
    for y in ys:
        c = np.zeros(span, np.complex64)
        for s0, l in zip(starts, lengths):
            c[s0 : s0 + l] = y[s0 : s0 + l]
        comps.append(c)
    d_true = [4321, 25000]
    for i, d0 in enumerate(d_true):
        rx[d0 : d0 + span] += (3 * comps[i] * np.exp(2j * np.pi * freqs[7 + 11 * i] * np.arange(span) / fs)).astype(np.complex64)
    tm = np.stack(comps)
    d_rx = asarray(rx)
    kw = dict(max_rx_len=m, freqs_norm=freqs / fs, group_starts=starts, group_lens=lengths)
    plan = CAFPlan(tm, **kw)
    assert plan.engine_used == "direct"
    S = m - span + 1
    res = plan.run(d_rx, surface=True)
    surf = res.surface.get()
    assert surf.shape == (T, S, freqs.size)
    rows = np.unique(np.concatenate((rng.integers(0, S, 300), d_true, [0, S - 1])))
    for i in range(T):
        ref = O.GroupXcorr(ys[i], starts, lengths, freqs, fs).caf(rx, rows)
        assert np.max(np.abs(surf[i][rows] - ref)) <= 3e-6 * max(1.0, float(ref.max()))
        np.testing.assert_array_equal(res.row_max.get()[i], surf[i].max(axis=1))
        np.testing.assert_array_equal(res.row_arg.get()[i], np.argmax(surf[i], axis=1))
        j = int(np.argmax(res.row_max.get()[i]))
        assert int(res.peak_delay.get()[i]) == j == d_true[i]
        assert float(res.peak_val.get()[i]) == res.row_max.get()[i][j] and int(res.peak_freq.get()[i]) == 7 + 11 * i
    # no surface: identical rows and peaks; a sub-range equals the slice
    r2 = plan.run(d_rx, surface=False)
    np.testing.assert_array_equal(r2.row_max.get(), res.row_max.get())
    np.testing.assert_array_equal(r2.row_arg.get(), res.row_arg.get())
    np.testing.assert_array_equal(r2.peak_delay.get(), res.peak_delay.get())
    r3 = plan.run(d_rx, shift_start=4000, num_shifts=1000, surface=True)
    np.testing.assert_array_equal(r3.surface.get(), surf[:, 4000:5000])
    assert int(r3.peak_delay.get()[0]) == d_true[0]
    # the overlap-save engine agrees at ITS tolerance for such a template
    per = CAFPlan(tm, engine="persistent", **kw).run(d_rx, surface=True)
    scale = float(np.nanmax(surf))
    assert np.nanmax(np.abs(per.surface.get() - surf)) <= 2e-5 * scale * 64.0 / lengths.sum()
    np.testing.assert_array_equal(per.peak_delay.get(), res.peak_delay.get())
    # explicit selection on a plain short template (bins mode), and the limit of 64 non-zero samples
    t48 = qpsk(rng, 48)
    rx48 = cn(rng, 5000)
    rx48[1234 : 1234 + 48] += (2 * t48 * np.exp(2j * np.pi * 5 * np.arange(48) / 48)).astype(np.complex64)
    b48 = np.arange(-8, 9)
    pd48 = CAFPlan(t48, max_rx_len=5000, bins=b48, grid=48, engine="direct")
    assert pd48.engine_used == "direct"
    r48 = pd48.run(asarray(rx48), surface=True)
    assert (int(r48.peak_delay.get()[0]), int(b48[r48.peak_freq.get()[0]])) == (1234, 5)
    ref48 = O.caf_bins(t48, rx48, b48, np.arange(5000 - 48 + 1))
    assert np.max(np.abs(r48.surface.get()[0] - ref48)) <= 3e-6 * float(ref48.max())
    with pytest.raises(ValueError, match="64 non-zero"):
        CAFPlan(qpsk(rng, 100), max_rx_len=5000, bins=[0], grid=100, engine="direct")
    with pytest.raises(ValueError):
        plan.run(d_rx, cqf=True)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,nb", [(1000, 40000, 5), (4096, 70000, 17), (8192, 50000, 3), (300, 3000, 1)])
def test_complex_qf_rows_from_the_persistent_engine(n, m, nb):
    """caf_outputs::d_cqf from the one-launch in-LDS engine (its FFT items write the complex rows themselves) against the
    rocFFT engine (multiply -> rocFFT -> normalise) and against the definition r / (||t|| ||window||) of
    fastXcorr(absResult=False) / TemplateCrossCorrelator (xcorrRoutines.py:533-548, :352-357); sub-ranges of delays and
    several templates x bins; the plane is the call's only output, anything else with it is refused."""
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(n + nb)
    T = 3
    tm = np.stack([qpsk(rng, n) * (0.5 + k) for k in range(T)])
    rx = cn(rng, m)
    rx[m // 3 : m // 3 + n] += tm[1]
    grid = 1 << int(np.ceil(np.log2(n)))
    bins = np.arange(-(nb // 2), nb - nb // 2)
    d_rx = asarray(rx)
    planes = {}
    for engine in ("persistent", "rocfft"):
        plan = CAFPlan(tm, max_rx_len=m, bins=bins, grid=grid, engine=engine)
        assert plan.engine_used == engine
        planes[engine] = plan.run(d_rx, rows=False, peak=False, cqf=True).cqf.get()
        lo, cnt = 777, min(1234, m - n + 1 - 777)
        sub = plan.run(d_rx, shift_start=lo, num_shifts=cnt, rows=False, peak=False, cqf=True).cqf.get()
        # (another alignment of the overlap-save blocks: equal to float32 rounding, not bit for bit)
        np.testing.assert_allclose(sub, planes[engine][:, :, lo : lo + cnt], atol=2e-6)
        if engine == "persistent":
            with pytest.raises(ValueError):
                plan.run(d_rx, cqf=True)  # rows / peaks together with the plane: the rocFFT engine's job
        plan.close()
    assert planes["persistent"].shape == (T, nb, m - n + 1)
    assert np.max(np.abs(planes["persistent"] - planes["rocfft"])) <= 2e-5
    # the definition, on a sample of delays: sum_k rx[d + k] conj(t[k]) e^{-j 2 pi b k / grid} / (||t|| ||rx[d : d + n]||)
    sel = np.unique(np.concatenate((rng.integers(0, m - n + 1, 40), [0, m - n, m // 3])))
    k = np.arange(n)
    for t in range(T):
        for bi, b in enumerate(bins):
            u = np.conj(tm[t].astype(np.complex128)) * np.exp(-2j * np.pi * b * k / grid)
            for d in sel:
                w = rx[d : d + n].astype(np.complex128)
                ref = np.dot(w, u) / (np.linalg.norm(tm[t].astype(np.complex128)) * np.linalg.norm(w))
                assert abs(planes["persistent"][t, bi, d] - ref) <= 2e-5
    assert abs(abs(planes["persistent"][1, nb // 2, m // 3]) - 1.0) < 0.3  # the planted copy (noise is as strong as it)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(1000, 60000), (4096, 90000), (8192, 70000)])
def test_no_frequency_scan_peaks_come_from_the_work_items(n, m):
    """F = 1 on the one-launch engine (config C3: GroupXcorrFFT-style banks, per-template argmax): the FFT items write the
    finished rows AND one peak record per (template, block, wave); the peak of a template must be the maximum of its own
    row and the FIRST delay that holds it, bit for bit -- with ties planted (two identical copies of a template, a
    periodic rx), with zero-energy windows (NaN rows never win), on sub-ranges, and in a call that asks for the peaks only."""
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(n)
    T = 5
    tm = np.stack([qpsk(rng, n) for _ in range(T)])
    rx = cn(rng, m)
    # template 1: two identical, noise-free copies -> QF^2 = 1 twice, the first one must be reported
    rx[5000 : 5000 + n] = tm[1]
    rx[5000 + 3 * n : 5000 + 4 * n] = tm[1]
    # template 2: planted inside a stretch of zeros (zero-energy windows around it: NaN rows)
    rx[30000 - 2 * n : 30000 + 3 * n] = 0
    rx[30000 : 30000 + n] = 0.5 * tm[2]
    d_rx = asarray(rx)
    plan = CAFPlan(tm, max_rx_len=m, bins=[0], grid=1 << int(np.ceil(np.log2(n))), engine="persistent")
    for lo, cnt in ((0, None), (123, 40000), (4999, 3 * n + 2)):
        kw = dict(shift_start=lo) if cnt is None else dict(shift_start=lo, num_shifts=cnt)
        res = plan.run(d_rx, rows=True, peak=True, **kw)
        rows = res.row_max.get()
        assert np.all(res.row_arg.get() == 0)
        pv, pd = res.peak_val.get(), res.peak_delay.get()
        for t in range(T):
            r = np.where(np.isnan(rows[t]), -1.0, rows[t])
            assert pv[t] == r.max(), (t, lo)
            assert pd[t] == lo + int(np.argmax(r)), (t, lo)  # first index of the maximum
        only = plan.run(d_rx, rows=False, peak=True, **kw)
        np.testing.assert_array_equal(only.peak_val.get(), pv)
        np.testing.assert_array_equal(only.peak_delay.get(), pd)
        assert np.all(only.peak_freq.get() == 0)
    res = plan.run(d_rx, rows=True, peak=True)
    assert res.peak_delay.get()[1] == 5000 and abs(res.peak_val.get()[1] - 1.0) < 1e-5
    assert res.peak_delay.get()[2] == 30000 and abs(res.peak_val.get()[2] - 1.0) < 1e-5
    assert np.isnan(res.row_max.get()[2][30000 - n - 5])  # a window of zeros: 0 / 0
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["persistent", "fused", "rocfft"])
def test_zero_energy_windows_are_nan_and_never_win(engine):
    """A stretch of exact zeros in rx longer than the template (a gap in a recording): the reference's 0 / 0.  Every engine
    must report NaN on the surface and in the per-delay maxima (hypothesis 0) for those windows -- not the overlap-save
    transform's rounding noise times 1 / 0 -- and must not let them near the peak."""
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(1)
    n, m = 1000, 40000
    t = qpsk(rng, n)
    rx = cn(rng, m)
    rx[20000:24000] = 0
    rx[9000 : 9000 + n] += t
    d_rx = asarray(rx)
    z = slice(20000, 24000 - n + 1)
    for F in (1, 8):
        plan = CAFPlan(t, max_rx_len=m, bins=np.arange(F) - F // 2, grid=1024, engine=engine)
        for surf in (True, False):
            res = plan.run(d_rx, surface=surf, rows=True, peak=True)
            rm, ra = res.row_max.get()[0], res.row_arg.get()[0]
            assert np.all(np.isnan(rm[z])) and np.all(ra[z] == 0), (F, surf)
            assert not np.any(np.isnan(rm[: 20000 - n])) and not np.any(np.isnan(rm[24000:]))
            if surf:
                assert np.all(np.isnan(res.surface.get()[0][z]))
            assert int(res.peak_delay.get()[0]) == 9000 and int(res.peak_freq.get()[0]) == F // 2
            assert res.peak_val.get()[0] == np.nanmax(rm)
        plan.close()


@pytest.mark.parametrize("T,F,n,m", [(1, 5, 128, 40000), (2, 32, 300, 30000), (3, 200, 64, 20000), (1, 256, 4096, 70000),
                                     (2, 64, 8192, 50000), (3, 2, 256, 30000), (5, 3, 100, 20000), (1, 130, 512, 30000)])
def test_hypothesis_major_surface_is_the_transposed_surface_bit_for_bit(T, F, n, m):
    """caf_outputs.d_surface_t (T, F, S): written by the FFT work items themselves (fused_item MODE 5); the same numbers as
    the delay-major surface the tile role writes, and row_max / row_arg / the peak are the maximum / first argmax of them."""
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(1000 * T + F)
    tm = np.stack([qpsk(rng, n) for _ in range(T)])
    rx = cn(rng, m)
    d0 = m // 3
    rx[d0 : d0 + n] += (2 * tm[T - 1] * np.exp(2j * np.pi * 2 * np.arange(n) / n)).astype(np.complex64)
    rx[m // 2 : m // 2 + 2 * n] = 0  # a stretch of zero-energy windows: NaN in both layouts
    bins = np.arange(F) - F // 2
    plan = CAFPlan(tm, max_rx_len=m, bins=bins, grid=n if 16384 % n == 0 else 16384, engine="persistent")
    d_rx = asarray(rx)
    ref = plan.run(d_rx, surface=True)
    got = plan.run(d_rx, surface_t=True)
    a, b = ref.surface.get(), got.surface_t.get()
    assert b.shape == (T, F, m - n + 1)
    np.testing.assert_array_equal(np.transpose(a, (0, 2, 1)), b)  # (NaN == NaN in assert_array_equal)
    assert np.isnan(b).any()
    np.testing.assert_array_equal(got.row_max.get(), ref.row_max.get())
    np.testing.assert_array_equal(got.row_arg.get(), ref.row_arg.get())
    for k in ("peak_val", "peak_delay", "peak_freq"):
        np.testing.assert_array_equal(getattr(got, k).get(), getattr(ref, k).get())
    # against the oracle on sampled delays of the last template
    sel = np.unique(np.concatenate((np.arange(0, 40), np.arange(d0 - 20, d0 + 20), np.arange(m - n + 1 - 40, m - n + 1))))
    want = O.caf_bins(tm[T - 1], rx, bins, sel) if 16384 % n == 0 else None
    if want is not None:
        assert np.max(np.abs(b[T - 1][:, sel].T - want)) <= 1e-4 * want.max()
    # a sub-range, the surface alone (no per-delay results, no peak)
    lo, cnt = 777, min(13000, m - n + 1 - 777)
    sub = plan.run(d_rx, shift_start=lo, num_shifts=cnt, surface_t=True, rows=False, peak=False)
    # (another run start = other block boundaries: equal up to the rounding of the transform, like every sub-range run)
    np.testing.assert_allclose(sub.surface_t.get(), b[:, :, lo : lo + cnt], atol=2e-6, equal_nan=True)
    assert plan.watchdog() == (0, 0)
    plan.close()


def test_hypothesis_major_surface_rules():
    """One hypothesis per template: the two layouts coincide (any engine); otherwise it is the persistent engine's output and
    excludes the delay-major surface and the complex plane."""
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(77)
    tm, rx = np.stack([qpsk(rng, 100) for _ in range(3)]), cn(rng, 9000)
    for engine in ("persistent", "rocfft"):
        plan = CAFPlan(tm, max_rx_len=rx.size, bins=[0], grid=1024, engine=engine)
        r = plan.run(asarray(rx), surface_t=True)
        ref = plan.run(asarray(rx), surface=True)
        np.testing.assert_array_equal(r.surface_t.get()[:, 0, :], ref.surface.get()[:, :, 0])
        np.testing.assert_array_equal(r.row_max.get(), ref.row_max.get())
        plan.close()
    plan = CAFPlan(tm, max_rx_len=rx.size, bins=[0, 1, 2], grid=1024, engine="rocfft")
    with pytest.raises(ValueError, match="hypothesis-major"):
        plan.run(asarray(rx), surface_t=True)
    plan.close()
    plan = CAFPlan(tm, max_rx_len=rx.size, bins=[0, 1, 2], grid=16384)
    with pytest.raises(ValueError, match="hypothesis-major"):
        plan.run(asarray(rx), surface_t=True, surface=True)
    plan.close()


@pytest.mark.parametrize("engine", ["persistent", "fused", "rocfft"])
def test_quiet_windows_late_in_a_long_record(engine):
    """10^7 loud samples, then a stretch 90 dB below them that holds a (scaled) copy of the template, then a gap of exact
    zeros.  On every engine the quiet windows are FINITE -- and right: QF^2 does not depend on the scale -- and the gap is NaN
    (only an energy of exactly zero is: csrc/caf_energy.h; the deeper levels: test_quiet_windows_down_to_minus_160_db)."""
    from pydsproutines_amd import CAFPlan, asarray

    rng = np.random.default_rng(9)
    n, loud, quiet_len = 4096, 10_000_000, 60000
    t = qpsk(rng, n)
    a = np.float32(10 ** (-90 / 20))
    rx = np.concatenate((cn(rng, loud), a * cn(rng, quiet_len), np.zeros(3 * n, np.complex64), cn(rng, 20000)))
    d0 = loud + 30000
    rx[d0 : d0 + n] += a * t
    m = rx.size
    bins = np.arange(-2, 2)
    plan = CAFPlan(t, max_rx_len=m, bins=bins, grid=n, engine=engine)
    # (the last 2^20 samples only for the rocfft engine's sake: the prefix still runs over the whole record)
    lo = loud - 50000
    res = plan.run(asarray(rx), shift_start=lo, num_shifts=m - n + 1 - lo, surface=True)
    rm = res.row_max.get()[0]
    quiet = slice(loud + 16384 - lo, loud + quiet_len - n - lo)       # windows wholly inside the quiet stretch
    gap = slice(loud + quiet_len - lo, loud + quiet_len + 2 * n - lo)  # windows wholly inside the zeros
    assert not np.any(np.isnan(rm[quiet])), "quiet windows (-90 dB behind 10^7 loud samples) must stay finite"
    assert np.all(np.isnan(rm[gap])) and np.all(res.row_arg.get()[0][gap] == 0)
    assert np.all(np.isnan(res.surface.get()[0][gap]))
    # the planted copy is found at full height although it is 90 dB down: the peak of the whole run
    assert int(res.peak_delay.get()[0]) == d0 and int(bins[res.peak_freq.get()[0]]) == 0
    assert 0.35 < float(res.peak_val.get()[0]) < 0.65
    sel = np.arange(d0 - 40, d0 + 40)
    ref = O.caf_bins(t, rx, bins, sel)
    # (tolerance: the window energy is 4e-13 of the float64 prefix it is a difference of on the rocfft engine -- resolved to
    #  about one part in a thousand; the 16384-point engines take it from a block-local prefix and hold 1e-4 here)
    assert np.max(np.abs(res.surface.get()[0][sel - lo] - ref)) <= (5e-3 if engine == "rocfft" else 1e-4) * ref.max()
    plan.close()


_QUIET_RX = {}


def _quiet_record(db):
    """2^24 unit-power samples, then -- `db` below them -- 3 x 65536 samples with a scaled copy of the template, exact zeros and
    20000 more quiet samples: from the start of the quiet stretch on, every engine's overlap-save blocks (and their float32
    transforms) see nothing but quiet samples."""
    if db not in _QUIET_RX:
        _QUIET_RX.clear()  # (one 134 MB record at a time)
        rng = np.random.default_rng(1000 - db)
        n, loud, quiet_len = 4096, 1 << 24, 3 * 65536
        t = qpsk(rng, n)
        a = np.float32(10.0 ** (db / 20))
        rx = np.concatenate((cn(rng, loud), a * cn(rng, quiet_len), np.zeros(3 * n, np.complex64), a * cn(rng, 20000)))
        d0 = loud + 30000
        rx[d0 : d0 + n] += a * t
        _QUIET_RX[db] = (t, rx, loud, quiet_len, d0)
    return _QUIET_RX[db]


@pytest.mark.parametrize("engine", ["persistent", "fused", "rocfft"])
@pytest.mark.parametrize("db", [-100, -120, -160])
def test_quiet_windows_down_to_minus_160_db(engine, db):
    """The reference divides by the window's own norm (xcorrRoutines.py:527-528; ippsNorm_L2 per window, IppXcorrFFT.cpp:133-175)
    and is finite for every window that is not exactly zero.  Windows 100 / 120 / 160 dB below the 2^24 unit-power samples in
    front of them: a float64 prefix over the record resolves 2^-53 x 2^24 x (a few) -- the -100 dB windows to two digits, the
    others not at all -- so their energies are summed again directly (caf_energy.h).  Finite, within 1e-4 of the oracle, and
    the same NaN pattern -- exactly the windows of zeros -- on every engine."""
    from pydsproutines_amd import CAFPlan, asarray

    t, rx, loud, quiet_len, d0 = _quiet_record(db)
    n, m = t.size, rx.size
    bins = np.arange(-2, 2)
    plan = CAFPlan(t, max_rx_len=m, bins=bins, grid=n, engine=engine)
    lo = loud  # block 0 starts with the quiet stretch
    S = m - n + 1 - lo
    res = plan.run(asarray(rx), shift_start=lo, num_shifts=S, surface=True)
    rm, surf = res.row_max.get()[0], res.surface.get()[0]
    # NaN <=> the window holds nothing but zeros
    zeros_lo, zeros_hi = quiet_len, quiet_len + 3 * n  # (relative to lo) the zeros are samples [zeros_lo, zeros_hi)
    expect_nan = np.zeros(S, bool)
    expect_nan[zeros_lo : zeros_hi - n + 1] = True
    assert np.array_equal(np.isnan(rm), expect_nan)
    assert np.array_equal(np.isnan(surf).all(axis=1), expect_nan) and np.array_equal(np.isnan(surf).any(axis=1), expect_nan)
    assert np.all(res.row_arg.get()[0][expect_nan] == 0)
    # right to 1e-4 (QF^2 does not depend on the scale): around the planted copy, anywhere in the quiet stretch, and across
    # the edges of the zeros (windows that hold a few quiet samples and zeros otherwise)
    rng = np.random.default_rng(3)
    sel = np.unique(np.concatenate((np.arange(d0 - lo - 40, d0 - lo + 40), rng.integers(0, quiet_len - n, 60),
                                    np.arange(zeros_lo - n + 1, zeros_lo - n + 9), np.arange(zeros_hi - 8, zeros_hi))))
    ref = O.caf_bins(t, rx, bins, sel + lo)
    assert 0.3 < ref.max() < 0.7
    assert np.max(np.abs(surf[sel] - ref)) <= 1e-4 * ref.max()
    assert int(res.peak_delay.get()[0]) == d0 and int(bins[res.peak_freq.get()[0]]) == 0
    plan.close()


@pytest.mark.parametrize("n", [1000, 1200, 1430, 1450, 4096])
def test_quiet_windows_on_the_per_delay_path(n):
    """The same record through fastXcorr's frequency-search branch: the radix-10 kernel (1000), the mixed-radix kernel (1200), the
    three-kernel form (1450 = 2 5^2 29: k_sliding_multiply), a length with factors 11 and 13 (1430) -- all of them normalise from the
    float64 prefix -- and the power-of-two kernel,
    which sums its own windows: -120 dB windows finite and within 1e-4 of the oracle, windows of zeros (NaN, 0)."""
    from pydsproutines_amd.xcorrRoutines import fastXcorr

    t, rx, loud, quiet_len, d0 = _quiet_record(-120)
    cut = rx[d0 : d0 + n].copy()
    sh = np.concatenate((np.arange(d0 - 20, d0 + 20), np.arange(loud + 100, loud + 140),
                         np.arange(loud + quiet_len, loud + quiet_len + 8)))  # ... and 8 windows of zeros (n <= 3 x 4096 - 8)
    q, fi = fastXcorr(cut, rx, freqsearch=True, shifts=sh)
    rq, rfi = O.fastXcorr(cut, rx, freqsearch=True, shifts=sh[:-8])
    assert np.all(np.isnan(q[-8:])) and np.all(fi[-8:] == 0)
    assert not np.any(np.isnan(q[:-8]))
    np.testing.assert_allclose(q[:-8], rq, atol=1e-4)
    assert abs(q[20] - 1.0) < 1e-4 and fi[20] == 0
    np.testing.assert_array_equal(fi[:40], rfi[:40])
