#!/usr/bin/env python3
"""Workloads for the per-kernel roofline evidence (scripts/gpu_profile_kernels.sh runs every workload under
rocprofv3 three times: kernel trace, FETCH_SIZE, WRITE_SIZE).

    python3 scripts/profile_workloads.py <workload> [manifest.json]

Each workload exercises a few hand-written kernels at ONE shape (the BASELINE configs and the reference's own
benchmark shapes) and declares, per kernel, the ALGORITHMIC bytes (SURVEY 8d's per-unit figure x the units one launch
processes) and/or flops of one launch; scripts/summarize_kernels.py divides them by the traced durations.
`list` prints the workload names."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd import CAFPlan, _lib, asarray  # noqa: E402
from pydsproutines_amd.devarray import empty  # noqa: E402

N, M = 4096, 1 << 24
S = M - N + 1
rng = np.random.default_rng(8)
lib = _lib.load()


def sync():
    _lib.check(lib.caf_stream_sync(None))


# The trace pass repeats a workload (PROFILE_REPS_SCALE).  Plans, inputs and result buffers of the big launches are kept
# across the repetitions: a 17 GB surface allocated afresh is touched for the first time by the dispatch that is being timed,
# and rocprofv3's AVERAGE over the dispatches then reads ~10 % above what a resident buffer gives (median 15.3 ms, minimum
# 13.7 ms for the C2 launch) -- which is not what bench.py, a pipeline or the kernel's roofline sees.
_KEEP = {}


def keep(key, make):
    if key not in _KEEP:
        _KEEP[key] = make()
    return _KEEP[key]


def c2_inputs():
    def make():
        t = qpsk(rng, N)
        rx = cn(rng, M)
        rx[5_000_000 : 5_000_000 + N] += (t * np.exp(2j * np.pi * 37 * np.arange(N) / N)).astype(np.complex64)
        return t, asarray(rx)

    return keep("c2_inputs", make)


def caf(engine, surface, T=1, F=256, reps=2, rows=True):
    t, d_rx = c2_inputs()
    key = ("caf", engine, surface, T, F, rows)
    bins = np.arange(-F // 2, F // 2) if F > 1 else [0]
    plan = keep(key + ("plan",), lambda: CAFPlan(t if T == 1 else np.stack([qpsk(rng, N) for _ in range(T)]), max_rx_len=M,
                                                 bins=bins, grid=N, engine=engine))
    st = keep(key + ("res",), dict)
    for _ in range(reps):
        st["res"] = plan.run(d_rx, surface=surface, rows=rows, peak=True, out=st.get("res"))
    sync()
    B, step, nb = plan.block, plan.step, plan.blocks_per_batch
    nblk = -(-S // step)
    tiles = -(-step // 64)
    cells = float(T) * F * S
    fft_flops = nblk * T * F * (5.0 * B * np.log2(B) + 6.0 * B + 3.0 * step)
    man = []
    if plan.engine_used == "persistent":
        if surface:
            alg = nblk * (tiles * 64 * T * F * 4.0 + 8.0 * B * (T * F / 64.0 + 1)) + cells * 8.0 + T * S * 12.0
        elif F >= 64:
            gpt = -(-F // 64)
            alg = nblk * (tiles * 64 * T * gpt * 8.0 * 2 + 8.0 * B * (T * F / 64.0 + 1)) + T * S * 12.0
        elif F == 1:  # the FFT items write the finished rows themselves (no tiles): block spectra read, rows written
            alg = nblk * 8.0 * B * (T / 64.0 + 1) + T * S * 4.0
        else:  # fewer frequencies than a group: tiles written + read, traces written
            alg = nblk * (tiles * 64 * T * F * 4.0 * 2 + 8.0 * B * (T * F / 64.0 + 1)) + T * S * 8.0
        man.append(("k_caf_persistent", "one-launch engine, T=%d F=%d %s" % (T, F, "surface" if surface else "no surface"),
                    alg, fft_flops, reps))
    elif plan.engine_used == "fused":
        man.append(("k_fused_caf", "multiply + LDS IFFT + |.|^2 tiles", nblk * (tiles * 64 * T * F * 4.0 + 8.0 * B * (T * F / 64.0 + 1)),
                    fft_flops, reps))
        man.append(("k_transpose_norm_argmax", "tiles -> surface + argmax", cells * (4.0 + (4.0 if surface else 0.0)) + T * S * 12.0, 0.0, reps))
    else:
        Bc = plan.block
        man.append(("k_spectral_mul", "X * conj(H_h) for all hypotheses (8(1+2/F) B per point)", nblk * 8.0 * Bc * (T * F + 2), 0.0, reps))
        man.append(("k_magsq_norm_argmax", "|.|^2 + normalise + argmax (8 B read + 4 B write per cell)", cells * (8.0 + (4.0 if surface else 0.0)) + T * S * 12.0, 0.0, reps))
    if plan.engine_used == "persistent" and F == 1 and not surface and os.environ.get("CAF_F1_ITEM_PEAKS") == "0":
        # (default: the FFT items leave the peak records themselves and this pass over the rows does not run)
        man.append(("k_rows_peak", "peak records from the finished rows (4 B read per template and delay)", T * S * 4.0, 0.0, reps))
    if plan.engine_used in ("persistent", "fused") and B == 16384:
        # gather + in-LDS forward transform + sliding energies in one launch: rx samples read once per block, spectra
        # and 1/energy written
        man.append(("k_block_spectra", "block spectra + sliding energies: gather, in-LDS forward FFT, f64 block prefix "
                    "(8 B read + 8 B written per block point, 4 B per delay)", nblk * B * 16.0 + S * 4.0, nblk * 5.0 * B * np.log2(B), reps))
    else:
        man.append(("k_prefix_tiles|k_scan_tile_sums", "f64 energy prefix of |rx|^2: tile totals, their scan, the write (8 B read + 8 B written per sample)", M * 16.0, 0.0, reps))
        man.append(("k_inv_energy", "1 / window energy (16 B read + 4 B written per delay)", S * 20.0, 0.0, reps))
        man.append(("k_gather_blocks", "overlap-save blocks (8 B read + 8 B written per block point)", nblk * B * 16.0, 0.0, reps))
    return man


def w_c2_surface():
    return caf("persistent", True)


def w_c2_nosurface():
    return caf("persistent", False)


def w_c2_fused():
    return caf("fused", True)


def w_c2_rocfft():
    return caf("rocfft", True, reps=1)


def w_c3():
    return caf("persistent", False, T=64, F=1)


def w_c3_complex_rows():
    """Config C3 through the reference's literal call: TemplateCrossCorrelator.correlate(x) -> complex64 (64, S): the FFT items
    of the one-launch engine write the complex rows themselves (8 B written per template and delay)."""
    from pydsproutines_amd.xcorrRoutines import TemplateCrossCorrelator

    T = 64
    tm = np.stack([qpsk(rng, N) for _ in range(T)])
    _, d_rx = c2_inputs()
    tcc = TemplateCrossCorrelator(asarray(tm), M)
    for _ in range(2):
        out = tcc.correlate(d_rx)
    sync()
    del out
    nblk = -(-S // tcc._plan.step)
    return [("k_caf_persistent", "one-launch engine, T=64 F=1, complex QF rows (8 B written per value, block spectra read once "
             "per 32 templates)", T * S * 8.0 + nblk * 8.0 * 16384 * (T / 32.0), nblk * T * (5.0 * 16384 * 14 + 6.0 * 16384 + 4.0 * tcc._plan.step), 2)]


def w_c3_complex_rows_12000():
    """The same call with 64 templates of 12000 samples: the plain chained role (32768-point blocks) writes the complex rows from its
    O halves (fused_item2q MODE 4; 8 B written per value; the window energies, block spectra and template rows are re-read from the L2)."""
    from pydsproutines_amd.xcorrRoutines import TemplateCrossCorrelator

    T, L = 64, 12000
    tm = keep("tcc12000_tm", lambda: np.stack([qpsk(rng, L) for _ in range(T)]))
    _, d_rx = c2_inputs()
    tcc = keep("tcc12000", lambda: TemplateCrossCorrelator(asarray(tm), M))
    for _ in range(2):
        out = tcc.correlate(d_rx)
    sync()
    del out
    assert tcc._plan.engine_used == "persistent" and tcc._plan.block == 32768
    Sn = M - L + 1
    nblk = -(-Sn // tcc._plan.step)
    return [("k_caf_persistent", "one-launch engine, T=64 F=1 N=12000 (B=32768 as 2 x 16384), complex QF rows",
             T * Sn * 8.0 + Sn * 4.0 + nblk * 8.0 * 32768 * (T / 32.0), nblk * T * (2 * 5.0 * 16384 * 14 + 6.0 * 32768 + 4.0 * tcc._plan.step), 2)]


def w_c4_share():
    return caf("persistent", False, T=64, F=512, reps=1, rows=False)


def w_c2_long_template():
    """C2 shape with a 16384-sample template: 32768-point blocks = two chained 16384-point in-LDS transforms."""
    n = 16384
    d_rx = keep("lt_rx", lambda: asarray(cn(rng, M)))
    plan = keep("lt_plan", lambda: CAFPlan(qpsk(rng, n), max_rx_len=M, bins=np.arange(-128, 128), grid=16384))
    st = keep("lt_res", dict)
    for _ in range(2):
        st["res"] = plan.run(d_rx, surface=True, out=st.get("res"))
    sync()
    Sn = M - n + 1
    B, step = plan.block, plan.step
    nblk = -(-Sn // step)
    tiles = -(-step // 64)
    # compulsory HBM bytes: |y|^2 tiles written + read, surface written, block spectra read once (their per-hypothesis
    # re-reads are L2 / Infinity-Cache traffic: visible in FETCH_SIZE, not algorithmic)
    alg = nblk * (tiles * 64 * 256 * 4.0 + 8.0 * B) + Sn * 256 * 8.0 + Sn * 12.0
    flops = nblk * 256 * (2 * 5.0 * 16384 * 14 + 6.0 * B + 10.0 * 16384 + 3.0 * step)
    return [("k_caf_persistent", "one-launch engine, N=16384 (B=32768 as 2 x 16384), F=256, surface", alg, flops, 2),
            ("k_block_spectra32", "32768-point block spectra, parity-major + butterfly order: gather, one DIF step, two in-LDS "
             "16384-point transforms (8 B read + 8 B written per point)", nblk * B * 16.0, nblk * (2 * 5.0 * 16384 * 14 + 10.0 * B), 2)]


def w_c2_surface_t():
    """C2 with the hypothesis-major surface (caf_outputs.d_surface_t): rows written by the FFT items, no tiles."""
    t, d_rx = c2_inputs()
    plan = keep("st_plan", lambda: CAFPlan(t, max_rx_len=M, bins=np.arange(-128, 128), grid=N))
    st = keep("st_res", dict)
    for _ in range(2):
        st["res"] = plan.run(d_rx, surface_t=True, out=st.get("res"))
    sync()
    B, step = plan.block, plan.step
    nblk = -(-S // step)
    # surface written once, (value, hypothesis) pairs per delay and group written + read, traces written, block spectra read
    alg = S * 256 * 4.0 + nblk * (-(-step // 64) * 64 * 4 * 8.0 * 2 + 8.0 * B * 5) + S * 12.0
    flops = nblk * 256 * (5.0 * B * np.log2(B) + 6.0 * B + 3.0 * step)
    return [("k_caf_persistent", "one-launch engine, hypothesis-major surface [F][S] written by the FFT items (no tiles)", alg, flops, 2)]


def w_c2_lb16():
    """C2 shape with a 32768-sample template: 65536-point blocks in the folded form = two chained 16384-point transforms per output residue."""
    n = 32768
    d_rx = keep("lb16_rx", lambda: asarray(cn(rng, M)))
    plan = keep("lb16_plan", lambda: CAFPlan(qpsk(rng, n), max_rx_len=M, bins=np.arange(-128, 128), grid=16384))
    st = keep("lb16_res", dict)
    for _ in range(2):
        st["res"] = plan.run(d_rx, surface=True, out=st.get("res"))
    sync()
    Sn = M - n + 1
    B, step = plan.block, plan.step
    nblk = -(-Sn // step)
    alg = nblk * ((step // 64) * 64 * 256 * 4.0 + 8.0 * B) + Sn * 256 * 8.0 + Sn * 12.0
    flops = nblk * 256 * 2 * (2 * 5.0 * 16384 * 14 + 17.0 * 32768 + 3.0 * 16384)  # per item: two transforms, the fold (2 products + sum [+ twiddle]), |y|^2
    return [("k_caf_persistent", "one-launch engine, N=32768 (B=65536 folded: 2 x 16384 per output residue), F=256, surface",
             alg, flops, 2)]


def w_c2_parts_65536():
    """C2 shape with a 65536-sample template: the folded 65536-point role with the template in two partitions of 32768 samples
    (products of the item's block and of the next one with the two partition spectra, summed before the transform)."""
    n = 65536
    d_rx = keep("lb16_rx", lambda: asarray(cn(rng, M)))
    plan = keep("parts_plan", lambda: CAFPlan(qpsk(rng, n), max_rx_len=M, bins=np.arange(-128, 128), grid=16384))
    assert plan.engine_used == "persistent" and plan.block == 65536
    st = keep("parts_res", dict)
    for _ in range(2):
        st["res"] = plan.run(d_rx, surface=True, out=st.get("res"))
    sync()
    Sn = M - n + 1
    B, step = plan.block, plan.step
    nblk = -(-Sn // step)
    alg = nblk * ((step // 64) * 64 * 256 * 4.0 + 8.0 * B) + Sn * 256 * 8.0 + Sn * 12.0
    flops = nblk * 256 * 2 * (2 * 5.0 * 16384 * 14 + 2 * 17.0 * 32768 + 3.0 * 16384)  # per item: two transforms, two folds, |y|^2
    return [("k_caf_persistent", "one-launch engine, N=65536 (B=65536 folded, template in 2 partitions of 32768), F=256, surface",
             alg, flops, 2)]


def w_c5_zoom():
    from pydsproutines_amd.zoom import caf_with_zoom

    t, d_rx = c2_inputs()
    bins = np.arange(-128, 128)
    plan = CAFPlan(t, max_rx_len=M, bins=bins, grid=N)
    res = plan.run(d_rx, surface=False, rows=True, peak=True)
    for _ in range(3):
        out = caf_with_zoom(plan, d_rx, res, bins, N, float(N), k=8, min_height=0.004, span_bins=1.0, step_bins=1.0 / 64)
    assert len(out) == 8
    nfft = 4320
    return [("k_local_max", "local maxima of the 2^24-delay trace, both launches (4 B read per delay)", S * 4.0, 0.0, 3),
            ("k_zoom_topk", "device top-k of the candidates", 0.0, 0.0, 3),
            ("k_zoom_rows", "8 product rows, rotated + pre-chirped + padded (8 B read x2, 8 B written)", 8 * (N * 16.0 + nfft * 8.0), 0.0, 3),
            ("k_rows_mul_vec", "CZT spectral / output chirp multiplies (16 B per element)", 8 * nfft * 16.0 + 8 * 129 * 16.0, 0.0, 3)]


def w_direct_small_support():
    """Composite template with 16 samples of support over a 3000-sample span, 64 explicit frequencies, 2^22-sample rx:
    the direct engine (what AUTO picks below 64 samples of support)."""
    m, span, F2 = 1 << 22, 3000, 64
    starts, lengths = np.array([0, 700, 2991]), np.array([5, 2, 9])
    comp = np.zeros(span, np.complex64)
    y = cn(rng, span)
    for a, l in zip(starts, lengths):
        comp[a : a + l] = y[a : a + l]
    d_rx = asarray(cn(rng, m))
    plan = CAFPlan(comp, max_rx_len=m, freqs_norm=np.linspace(-0.04, 0.04, F2), group_starts=starts, group_lens=lengths)
    assert plan.engine_used == "direct"
    res = None
    for _ in range(3):
        res = plan.run(d_rx, surface=True, rows=True, peak=True, out=res)
    sync()
    s2, k = m - span + 1, int(lengths.sum())
    plan.close()
    return [("k_direct_caf", "direct engine: 16 products per (delay, hypothesis), 64 hypotheses, surface written "
             "(8 B read per sample, 4 B written per cell + 8 B per delay)", m * 8.0 + s2 * (F2 * 4.0 + 8.0), s2 * F2 * (8.0 * k + 3), 3)]


def perdelay(n, num, label):
    rx = cn(rng, n + num)
    d_rx, d_cut = asarray(rx), asarray(rx[500 : 500 + n].conj().copy())
    q, fi = empty(num, np.float32), empty(num, np.int32)
    import ctypes as ct

    for _ in range(2):
        _lib.check(lib.caf_xcorr_perdelay(ct.c_void_p(d_cut.ptr), n, ct.c_void_p(d_rx.ptr), rx.size, 0, 1, num, 0,
                                          ct.c_void_p(q.ptr), ct.c_void_p(fi.ptr), None, None, 0, None))
    sync()
    return n, num


# The fused per-delay kernels keep the (rows, N) product matrix in LDS: the bytes they MOVE are the rx samples (read once
# from HBM, overlapping windows come back from the L2) and 8 B of results per row -- that is what is declared, and the bound
# they sit on is the f32 rate (5 N log2 N flop per row), not the HBM.
def _perdelay_bytes(n, num):
    return (n + num) * 8.0 + num * 8.0


def w_perdelay_fused_4096():
    n, num = perdelay(4096, 1_000_000, "fused")
    return [("k_perdelay_fused", "product -> LDS FFT -> |.|^2 -> argmax, N=4096 x 1e6 delays (f32-bound: no bytes scale with rows x N)",
             _perdelay_bytes(n, num), num * 5.0 * n * np.log2(n), 2)]


def w_perdelay_fused_256():
    n, num = perdelay(256, 1_000_000, "fused")
    return [("k_perdelay_fused", "N=256 x 1e6 delays (same accounting)", _perdelay_bytes(n, num), num * 5.0 * n * np.log2(n), 2)]


def w_perdelay_rows_10007():
    """A prime cutout length beyond what Bluestein's convolution fits in LDS (10007 > 10000): product rows -> rocFFT rows -> argmax."""
    n, num = perdelay(10007, 20_000, "rows")
    return [("k_sliding_multiply", "normalised product rows, N=10007 x 2e4 (8 B written per element)", num * n * 8.0, 0.0, 2),
            ("k_rows_argmax", "|.|^2 + first argmax per row (8 B read per element)", num * n * 8.0, 0.0, 2)]


def w_perdelay_bluestein_1450():
    """1450 = 2 5^2 29: Bluestein's chirp transform in the LDS image (two 3072-point transforms per row; flops counted as the
    cutout's own 5 N log2 N, as for every other length)."""
    n, num = perdelay(1450, 100_000, "bluestein")
    return [("k_pdj", "per-delay correlator for N=1450 (Bluestein, convolution length 3072), %d rows" % num, _perdelay_bytes(n, num),
             num * 5.0 * n * np.log2(n), 2)]


def w_perdelay_split_65536():
    """65536-sample cutouts: four residues of a 16384-point transform per row; every residue re-reads cutout and window from the L2
    (16 B per sample and residue = 4.2 MB per row, declared as bytes: the L2 -> CU delivery is what bounds this form)."""
    n, num = perdelay(65536, 20_000, "split")
    return [("k_pdj", "per-delay correlator for N=65536 (4 residues x 16384 points), %d rows; bytes = L2 reads of the products' operands" % num,
             num * 4.0 * n * 16.0, num * 5.0 * n * np.log2(n), 2)]


def w_perdelay_decimal_1000():
    """benchmark_xcorrs.py's default cutout (1000 samples): the per-delay algorithm on radix-10 passes in LDS."""
    n, num = perdelay(1000, 1_000_000, "fused")
    return [("k_perdelay_r10", "fused per-delay correlator, N=1000 x 1e6 rows (f32-bound; 5 N log2 N flop per row)",
             _perdelay_bytes(n, num), num * 5.0 * n * np.log2(n), 2)]


def _w_perdelay_jit(n):
    num = 100_000
    perdelay(n, num, "mixed")
    return [("k_pdj", "per-delay correlator compiled for N=%d at run time (caf_perdelay_jit.h), %d rows (f32-bound; 5 N log2 N flop per row)" % (n, num),
             _perdelay_bytes(n, num), num * 5.0 * n * np.log2(n), 2)]


def w_perdelay_mixed_1200():
    """2^a 3^b 5^c 7^d cutouts: the kernel compiled for the length at run time (caf_jit.hip); one workload per length (every length's
    kernel is called k_pdj)."""
    return _w_perdelay_jit(1200)


def w_perdelay_mixed_1400():
    return _w_perdelay_jit(1400)


def w_perdelay_mixed_5000():
    return _w_perdelay_jit(5000)


def w_cp_fastxcorr_1e7():
    from pydsproutines_amd.xcorrRoutines import cp_fastXcorr

    n = 10_000_000
    rx = cn(rng, n + 1000)
    cut = rx[300 : 300 + n].copy()
    d_rx = asarray(rx)
    sh = np.arange(236, 364)
    cp_fastXcorr(cut, d_rx, shifts=sh)
    cp_fastXcorr(cut, d_rx, shifts=sh)
    sync()
    return [("k_sliding_multiply", "128 product rows of 1e7 samples (8 B written per element)", 128 * n * 8.0, 0.0, 2),
            ("k_rows_argmax", "128 rows of 1e7 (8 B read per element)", 128 * n * 8.0, 0.0, 2)]


def w_kernels_misc():
    import ctypes as ct

    from pydsproutines_amd.cupyExtensions import cupyComplexMagnSq, cupyFindLocalMaxima, multiTemplateSlidingDotProduct
    from pydsproutines_amd.filterRoutines import CupyKernelFilter, cupyMovingAverage
    from pydsproutines_amd.usrpRoutines import Iq16FrontEnd, iq16_to_complex64

    man = []
    n = 1 << 24
    x = cn(rng, n)
    d_x = asarray(x)
    for _ in range(3):
        cupyComplexMagnSq(d_x, np.float32)
    man.append(("k_magnsq", "complex |.|^2 c64 -> f32 (12 B per element)", n * 12.0, 0.0, 4))
    d_p = cupyComplexMagnSq(d_x, np.float32)
    for _ in range(3):
        cupyMovingAverage(d_p, 100)
    man.append(("k_moving_tile", "moving average L=100 (4 B read + 4 B written per sample)", n * 8.0, 0.0, 3))
    d_t = asarray(cn(rng, 20 * 100).reshape(20, 100))
    for _ in range(2):
        multiTemplateSlidingDotProduct(d_x[: 10_000_100], d_t, 0, 10_000_000)
    man.append(("k_multi_template_dot", "20 templates x 100 over 1e7 slides (8 B read + 8 B written per slide; 2*20*100*8 flop per slide)",
                10_000_000 * 16.0, 10_000_000 * 20 * 100 * 8.0, 2))
    man.append(("k_prefix_tiles|k_scan_tile_sums", "float64 energy prefix of 1e7 samples: tile totals, their scan, the write (8 B read + 8 B written per sample)",
                10_000_100 * 16.0, 0.0, 2))
    f = CupyKernelFilter()
    d_tp = asarray((rng.standard_normal(1024) / 32).astype(np.float32))
    for _ in range(3):
        f.filter_smtaps(d_x, d_tp)
    man.append(("k_fir_os", "overlap-save FIR, 1024 taps, fused 4096-point blocks (8 B read + 8 B written per output)", n * 16.0,
                n / 3073.0 * 2 * 5 * 4096 * 12, 3))
    iq = rng.integers(-2000, 2000, 2 * n).astype(np.int16)
    d_iq = asarray(iq)
    for _ in range(3):
        iq16_to_complex64(d_iq, 1.0 / 2048)
    man.append(("k_iq16_to_c64", "int16 IQ -> complex64 (4 B read + 8 B written per sample)", n * 12.0, 0.0, 3))
    fe = Iq16FrontEnd(asarray((rng.standard_normal(64) / 8).astype(np.float32)), dsr=4, scale=1.0 / 2048)
    for _ in range(3):
        fe.reset()
        fe.run(d_iq)
    man.append(("k_fir_poly", "int16 IQ -> 64-tap FIR -> /4 (4 B read per input + 8 B written per output)", n * 4.0 + n / 4 * 8.0, n / 4 * 64 * 4.0, 3))
    d_m = asarray(cn(rng, 64 * (1 << 18)).reshape(64, 1 << 18))
    d_tp = asarray((rng.standard_normal(128) / 11).astype(np.float32))
    for _ in range(3):
        out = f.upfirdn_sm(d_m, d_tp, 5, 2)
    man.append(("k_upfirdn_poly|k_upfirdn", "upfirdn 64 x 2^18, 128 taps, up 5 down 2 (8 B read per input + 8 B written per output)",
                d_m.size * 8.0 + out.size * 8.0, 0.0, 3))
    for _ in range(3):
        cupyFindLocalMaxima(d_p, 3.0)
    man.append(("k_local_max", "local maxima of a 2^24 float trace, both launches (4 B read per sample)", n * 4.0, 0.0, 3))
    sync()
    return man


def w_fir_direct():
    """Direct-form FIR kernels (the dispatch would use the overlap-save form from 96 taps on: pinned by the A/B switch,
    which is read once per process)."""
    import os

    os.environ["CAF_FIR_OS_MIN_TAPS"] = "1000000"
    from pydsproutines_amd.filterRoutines import CupyKernelFilter

    n = 1 << 24
    d_x = asarray(cn(rng, n))
    f = CupyKernelFilter()
    man = []
    for taps in (64, 128):
        d_tp = asarray((rng.standard_normal(taps) / np.sqrt(taps)).astype(np.float32))
        for _ in range(3):
            f.filter_smtaps(d_x, d_tp)
    man.append(("k_fir_fast", "direct FIR, 64 and 128 taps, 2^24 samples (16 B per sample; 4 flop per tap)", n * 16.0 * 2,
                n * (64 + 128) * 4.0, 3))
    sync()
    return man


WORKLOADS = {k[2:]: v for k, v in list(globals().items()) if k.startswith("w_")}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] == "list":
        print(" ".join(WORKLOADS))
        sys.exit(0)
    name = sys.argv[1]
    # PROFILE_REPS_SCALE=k: the workload is run k times in this process (>= 10 dispatches of its dominant kernel for the
    # timing statistics of the trace pass; the counter passes keep k = 1, traffic per call does not vary)
    import os

    scale = max(1, int(os.environ.get("PROFILE_REPS_SCALE", "1")))
    for _ in range(scale):
        man = WORKLOADS[name]()
        sync()
    out = [{"kernel": k, "what": w, "alg_bytes_per_call": b, "alg_flops_per_call": fl, "calls": c * scale} for k, w, b, fl, c in man]
    if len(sys.argv) > 2:
        json.dump({"workload": name, "kernels": out}, open(sys.argv[2], "w"), indent=1)
    print("workload %s done: %d kernels declared" % (name, len(out)))
