"""The reference's benchmark scripts as call SEQUENCES (layer L4): the same functions, in the same order,
with the scripts' own arguments and cross-checks (verifyRoutines.compareValues between implementations), at sizes
that run in seconds.  Written from the cited call lines, not from the scripts' text; `cp.asarray` becomes this package's
`asarray` (there is no cupy on the box -- the one difference a user of the reference has to make).

    benchmarks/benchmark_xcorrs.py:28-59                      fastXcorr <-> CyIppXcorrFFT <-> cp_fastXcorr
    benchmarks/benchmark_groupXcorrs.py:37-72                 GroupXcorrCZT <-> pbIppGroupXcorrCZT (1 thread, 4 threads)
    benchmarks/benchmark_cupyTemplateCrossCorrelator.py:32-37 TemplateCrossCorrelator(...).correlate(dx, returnMax=True)
    benchmarks/benchmark_czts.py:31-82                        CZTCachedGPU / CZTCached / pbIppCZT32fc / dot-tones kernel
and the five kernel-level scripts:
    benchmarks/benchmark_xcorrKernels.py:25-76                cp_fastXcorr_v2 with its tuning arguments, twice (warm-up + timed)
    benchmarks/benchmark_multiTemplateDotKernels.py:24-64     randPSKsyms -> addManySigToNoise -> multiTemplateSlidingDotProduct
    benchmarks/benchmark_filterkernels.py:33-75               lfilter <-> filter_smtaps <-> filter_smtaps_sminput (128 / 1024 per block)
    benchmarks/benchmark_upfirdnkernels.py:15-67              upfirdn_naive per row <-> upfirdn_sm <-> scipy.signal.upfirdn, frac < 1e-4
    benchmarks/benchmark_movAvgKernels.py:19-48               filter_smtaps_sminput(ones / L) <-> cupyMovingAverage
"""

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def test_benchmark_xcorrs_sequence():
    """benchmark_xcorrs.py:28-59 with cutoutlen 1000 (its default), cupybatchsize 1 and 16, numShifts 128; the data is
    10^6 samples instead of 10^8 (only the first shifts are searched, as in the script)."""
    from pydsproutines_amd import asarray
    from pydsproutines_amd.timingRoutines import Timer
    from pydsproutines_amd.verifyRoutines import compareValues
    from pydsproutines_amd.xcorrRoutines import CyIppXcorrFFT, cp_fastXcorr, fastXcorr

    rng = np.random.default_rng(0)
    datalen, cutoutlen, numShifts = 1_000_000, 1000, 128
    x = (rng.standard_normal(datalen) + 1j * rng.standard_normal(datalen)).astype(np.complex64)
    start = 10000
    cutout = x[start : start + cutoutlen]
    timer = Timer()
    startIdx, endIdx, idxStep = 0, numShifts, 1
    shifts = np.arange(startIdx, endIdx, idxStep)
    timer.start()
    out = fastXcorr(cutout, x, freqsearch=True, shifts=shifts)
    timer.evt("cpu-signature fastXcorr")
    numThreads = 4
    cyxc = CyIppXcorrFFT(cutout, numThreads)
    cyout = cyxc.xcorr(x, startIdx, endIdx, idxStep)
    timer.evt("CyIppXcorrFFT, %d threads" % numThreads)
    d_cutout, d_x = asarray(cutout), asarray(x)
    for batch in (1, 16):
        cpout = cp_fastXcorr(d_cutout, d_x, freqsearch=True, shifts=shifts, BATCH=batch)
        timer.evt("cp_fastXcorr BATCH=%d" % batch)
        # the script's four comparisons (it prints them; here they are bounded)
        for a, b in ((out[0], cyout[0]), (out[0], cpout[0])):
            raw, frac = compareValues(a, b, verbose=False)
            assert raw <= 2e-5 and frac <= 5e-3
        np.testing.assert_array_equal(out[1], cyout[1])
        np.testing.assert_array_equal(out[1], cpout[1])
    timer.end()
    # return types of the three implementations (xcorrRoutines.py:511-531, CyIppXcorrFFT.pyx:64-78, xcorrRoutines.py:147-158)
    assert out[0].dtype == np.float64 and out[1].dtype == np.uint32
    assert cyout[0].dtype == np.float32 and cyout[1].dtype == np.int32
    assert cpout[0].dtype == np.float64 and cpout[1].dtype == np.uint32
    # and what they must say: no planted offset in the first 128 shifts, so they all agree with the oracle's branch B
    rq, rf = O.fastXcorr(cutout, x, freqsearch=True, shifts=shifts)
    assert np.max(np.abs(out[0] - rq)) <= 2e-5
    # a second run that DOES cover the cutout's position: QF^2 = 1 at shift `start`, bin 0
    sh2 = np.arange(start - 64, start + 64)
    o2 = fastXcorr(cutout, x, freqsearch=True, shifts=sh2)
    c2 = cyxc.xcorr(x, start - 64, start + 64, 1)
    p2 = cp_fastXcorr(d_cutout, d_x, freqsearch=True, shifts=sh2, BATCH=32)
    for q, f in (o2, c2, p2):
        assert int(np.argmax(q)) == 64 and abs(float(q[64]) - 1.0) < 1e-5 and int(f[64]) == 0


def test_benchmark_group_xcorrs_sequence():
    """benchmark_groupXcorrs.py:19-72: QPSK symbols, groups of `groupLength` every 2 groupLength samples from sample 100,
    CZT grid -100 .. 100 Hz in 1 Hz steps at fs = 10 kHz, 41 shifts around the first group; python GroupXcorrCZT against
    the pybind twin with its default thread count and with 4 threads.  100 000 symbols and groups of 500 instead of 10^6
    and 5000 (same structure: 100 groups)."""
    from pydsproutines_amd.signalCreationRoutines import randPSKsyms
    from pydsproutines_amd.timingRoutines import Timer
    from pydsproutines_amd.verifyRoutines import compareValues
    from pydsproutines_amd.xcorrRoutines import GroupXcorrCZT, pbIppGroupXcorrCZT

    timer = Timer()
    np.random.seed(5)
    x, _ = randPSKsyms(100000, 4, dtype=np.complex64)
    f1, f2, fstep, fs = -100.0, 100.0, 1.0, 10000
    firstGroupStart, groupLength = 100, 500
    groupStarts = np.arange(firstGroupStart, x.size, groupLength * 2, dtype=np.int32)
    assert groupStarts.size == 100
    timer.start()
    gxc = GroupXcorrCZT(x, groupStarts, np.zeros(groupStarts.size, dtype=np.int32) + groupLength, f1, f2, fstep, fs)
    timer.evt("preparing the python object")
    shiftStart, shiftStep, numShifts = firstGroupStart - 20, 1, 41
    results, cztfreq = gxc.xcorr(x, np.arange(shiftStart, shiftStart + numShifts, shiftStep))
    timer.evt("python object")
    assert results.shape == (numShifts, 201) and cztfreq.size == 201
    for threads in (None, 4):
        pbgxc = pbIppGroupXcorrCZT(groupLength, f1, f2, fstep, fs) if threads is None else pbIppGroupXcorrCZT(groupLength, f1, f2, fstep, fs, 4)
        assert pbgxc.getNumThreads() == (1 if threads is None else 4)
        for gs in groupStarts:
            pbgxc.addGroup(gs - firstGroupStart, x[gs : gs + groupLength])
        pbresults = pbgxc.xcorr(x, shiftStart, shiftStep, numShifts)
        timer.evt("pybind twin, %d thread(s)" % pbgxc.getNumThreads())
        assert pbresults.shape == results.shape and pbresults.dtype == np.float32
        raw, frac = compareValues(results.flatten(), pbresults.flatten(), verbose=False)
        assert raw <= 2e-5
    timer.end()
    # the template is a copy of the data: QF^2 = 1 at the first group's own position and 0 Hz
    assert abs(results[20, 100] - 1.0) < 1e-4 and np.unravel_index(np.argmax(results), results.shape) == (20, 100)
    ref = O.GroupXcorrCZT(x, groupStarts, np.zeros(groupStarts.size, np.int32) + groupLength, f1, f2, fstep, fs).xcorr(
        x, np.arange(shiftStart, shiftStart + numShifts, shiftStep))[0]
    assert np.max(np.abs(results - ref)) <= 2e-5


def test_benchmark_template_cross_correlator_sequence():
    """benchmark_cupyTemplateCrossCorrelator.py:19-37: cutouts of one QPSK signal at a regular jump, returnMax=True,
    called four times; every cutout must win its own delay with QF = 1."""
    from pydsproutines_amd import asarray
    from pydsproutines_amd.signalCreationRoutines import randPSKsyms
    from pydsproutines_amd.xcorrRoutines import TemplateCrossCorrelator

    np.random.seed(6)
    length, cutoutlen, cutoutstart, cutoutjump, numCutouts = 200000, 1000, 5000, 7000, 8
    x, _ = randPSKsyms(length, 4, dtype=np.complex64)
    cutouts = np.zeros((numCutouts, cutoutlen), dtype=x.dtype)
    for i in range(numCutouts):
        cutouts[i] = x[cutoutstart + cutoutjump * i : cutoutstart + cutoutjump * i + cutoutlen]
    dx, dcutouts = asarray(x), asarray(cutouts)
    correlator = TemplateCrossCorrelator(dcutouts, dx.size)
    out, ti = correlator.correlate(dx, returnMax=True)
    for _ in range(3):
        out, ti = correlator.correlate(dx, returnMax=True)
    o, t = out.get(), ti.get()
    assert o.shape == (length - cutoutlen + 1,) and o.dtype == np.float32 and t.dtype == np.int64
    for i in range(numCutouts):
        d = cutoutstart + cutoutjump * i
        assert abs(o[d] - 1.0) < 1e-5 and t[d] == i
    # returnMax is the column maximum of the complex output, bit for bit (the reference's unit-test property)
    # (|z| = the correctly rounded float32 magnitude, as in test_kat4_template_cross_correlator)
    full = correlator.correlate(dx).get()
    mag = np.sqrt(full.real.astype(np.float64) ** 2 + full.imag.astype(np.float64) ** 2).astype(np.float32)
    np.testing.assert_array_equal(o, mag.max(axis=0))
    np.testing.assert_array_equal(t, np.argmax(mag, axis=0))
    oq, oi = O.TemplateCrossCorrelator(cutouts, length).correlate(x, returnMax=True)
    assert np.max(np.abs(o - oq)) <= 2e-5


def test_benchmark_czts_sequence():
    """benchmark_czts.py:20-82: ten noise rows of length 10000, CZT over -1000 .. 1000 Hz in 1 Hz steps at fs = length:
    device object (runMany, run), the dot-tones kernel as a CZT without FFTs, the host object, the pybind twin (run in a
    python loop, runMany) -- all against each other (the script's commented-out check asks for fracChg < 1e-2)."""
    from pydsproutines_amd import asarray
    from pydsproutines_amd.signalCreationRoutines import randnoise
    from pydsproutines_amd.spectralRoutines import CZTCached, CZTCachedGPU, cupyDotTonesScaling, pbIppCZT32fc
    from pydsproutines_amd.verifyRoutines import compareValues

    np.random.seed(7)
    length, f1, f2, fstep = 10000, -1000.0, 1000.0, 1.0
    fs = length
    x = np.vstack([randnoise(length, 1, 1, 10).astype(np.complex64) for _ in range(10)])
    d_cztobj = CZTCachedGPU(length, f1, f2, fstep, fs)
    d_x = asarray(x)
    d_out = d_cztobj.runMany(d_x)
    d_single = d_cztobj.run(d_x[0])
    assert d_out.shape == (10, 2001) and d_single.shape == (2001,)
    d_inter = cupyDotTonesScaling(-f1 / fs, -fstep / fs, d_cztobj.getFreq().size, d_x[0])
    outkernel = d_inter.get().sum(axis=0)
    cztobj = CZTCached(length, f1, f2, fstep, fs, True)
    out = cztobj.runMany(x)
    pbczt = pbIppCZT32fc(length, f1, f2, fstep, float(length))
    pbout = np.stack([pbczt.run(x[i, :]) for i in range(x.shape[0])])
    pboutl = pbczt.runMany(x)
    ref = O.CZTCached(length, f1, f2, fstep, fs).runMany(x.astype(np.complex128))  # float64 constants and arithmetic
    scale = np.abs(ref).max()
    for name, got in (("device runMany", d_out.get()), ("host object", out), ("pybind loop", pbout), ("pybind runMany", pboutl)):
        raw, frac = compareValues(ref.flatten(), got.flatten(), verbose=False)
        assert raw <= 2e-5 * scale, name
    assert np.max(np.abs(d_single.get() - d_out.get()[0])) <= 1e-5 * scale
    assert np.max(np.abs(outkernel - ref[0])) <= 1e-4 * scale  # 10000-term float32 sums per bin
    raw, frac = compareValues(d_out.get().flatten(), out.flatten(), verbose=False)
    assert raw <= 2e-5 * scale


def test_benchmark_xcorr_kernels_sequence():
    """benchmark_xcorrKernels.py:25-76: cp_fastXcorr_v2(d_cutout, d_x, 0, numShifts, THREADS_PER_BLOCK, numSlidesPerBlk) with
    the script's defaults (cutoutlen 1000, 128 threads, 1024 slides per block) and one of its other settings, run twice
    as the script does (warm-up, then the timed call); 20000 shifts instead of 100000.  The CUDA tuning arguments are
    accepted and change nothing.  Note the script's convention: the cutout is x[:cutoutlen], NOT conjugated (v2 multiplies as
    given, xcorrRoutines.py:169-274) -- the oracle is asked for exactly that product."""
    from pydsproutines_amd import asarray
    from pydsproutines_amd.timingRoutines import Timer
    from pydsproutines_amd.xcorrRoutines import cp_fastXcorr_v2

    rng = np.random.default_rng(3)
    timer = Timer()
    for cutoutlen, tpb, spb in ((1000, 128, 1024), (1000, 256, 512), (1200, 128, 1024), (2048, 128, 3872)):
        numShifts = 20000
        datalen = cutoutlen + numShifts - 1
        x = (rng.standard_normal(datalen) + 1j * rng.standard_normal(datalen)).astype(np.complex64)
        cutout = x[:cutoutlen]
        d_cutout, d_x = asarray(cutout), asarray(x)
        cpout2 = cp_fastXcorr_v2(d_cutout, d_x, 0, numShifts, tpb, spb)  # warm-up
        timer.start()
        cpout2 = cp_fastXcorr_v2(d_cutout, d_x, 0, numShifts, tpb, spb)
        timer.end("cp_fastxcorr v2")
        # as the script calls it (flattenCAF left at False) the result is the float32 (numShifts, cutoutlen) |.|^2 plane
        assert cpout2.dtype == np.float32 and cpout2.shape == (numShifts, cutoutlen)
        # v2 multiplies by the cutout AS GIVEN: the oracle's branch C with conj(cutout) as its cutout (it conjugates again)
        sh = np.concatenate((np.arange(0, 48), rng.integers(0, numShifts, 48)))
        rows = O.fastXcorr(cutout.conj(), x, freqsearch=True, outputCAF=True, shifts=sh)
        got = np.stack([cpout2[int(r)].get() for r in sh])
        assert np.max(np.abs(got - rows)) <= 2e-5
        # the flattened form of the same call (one fused kernel for these lengths): (uint32 bin, float32 QF^2) per shift
        fi, q = cp_fastXcorr_v2(d_cutout, d_x, 0, numShifts, tpb, spb, flattenCAF=True)
        fi, q = fi.get(), q.get()
        assert fi.dtype == np.uint32 and q.dtype == np.float32 and fi.shape == q.shape == (numShifts,)
        assert np.max(np.abs(q[sh] - rows.max(axis=1))) <= 2e-5
        top2 = np.sort(rows, axis=1)[:, -2:]
        clear = top2[:, 1] - top2[:, 0] > 4e-5  # (ties in noise: the index is compared only where the margin is clear)
        np.testing.assert_array_equal(fi[sh][clear], np.argmax(rows, axis=1)[clear])


def test_benchmark_multi_template_dot_kernels_sequence():
    """benchmark_multiTemplateDotKernels.py:24-64 with its defaults (10 QPSK templates of 100 symbols at starts 0, 200, ...,
    10 dB, idxlen 100000 -> 30000 here) and its plotted expectation: template i wins at its own start with QF^2 near
    snr / (snr + 1); everything against the oracle's kernel restatement."""
    from oracle import kernels as K
    from pydsproutines_amd import asarray
    from pydsproutines_amd.cupyExtensions import multiTemplateSlidingDotProduct
    from pydsproutines_amd.signalCreationRoutines import addManySigToNoise, randPSKsyms

    np.random.seed(5)
    idxlen, numTemplates, templateLength = 30000, 10, 100
    templates = np.zeros((numTemplates, templateLength), dtype=np.complex64)
    for i in range(numTemplates):
        syms, bits = randPSKsyms(templateLength, 4, dtype=np.complex64)
        templates[i, :] = syms
    sigStartIdxList = np.arange(numTemplates) * 2 * templateLength
    _, rx = addManySigToNoise(idxlen + templateLength - 1, sigStartIdxList, templates, 1, 1, np.zeros(numTemplates) + 10)
    d_x = asarray(rx.astype(np.complex64))
    d_templates = asarray(templates)
    for tpb, spb in ((32, 100), (128, 100)):  # the function default and the script's command-line default
        d_templateIdx, d_qf2 = multiTemplateSlidingDotProduct(d_x, d_templates.conj(), 0, idxlen, numSlidesPerBlk=spb,
                                                              THREADS_PER_BLOCK=tpb)
        tidx, qf2 = d_templateIdx.get(), d_qf2.get()
        assert tidx.shape == qf2.shape == (idxlen,)
        for i, s0 in enumerate(sigStartIdxList):
            assert tidx[s0] == i and 0.8 < qf2[s0] < 1.0  # 10 dB: snr / (snr + 1) = 0.91
        rt, rq = K.multiTemplateSlidingDotProduct(rx.astype(np.complex64), templates.conj(), 0, idxlen)
        assert np.max(np.abs(qf2 - rq)) <= 2e-5
        strong = rq > 0.2  # (in noise-only slides two templates can tie to float32 rounding)
        np.testing.assert_array_equal(tidx[strong], rt[strong])


def test_benchmark_filter_kernels_sequence():
    """benchmark_filterkernels.py:33-75: 10^6 complex64 noise samples, firwin(128, 0.5): scipy lfilter, filter_smtaps,
    filter_smtaps_sminput with 128 (default) and 1024 outputs per block, compared with compareValues as the script does (it
    prints; here bounded), plus the decimating call the script keeps commented out (:78-85)."""
    import scipy.signal as sps

    from pydsproutines_amd import asarray
    from pydsproutines_amd.filterRoutines import CupyKernelFilter
    from pydsproutines_amd.signalCreationRoutines import randnoise
    from pydsproutines_amd.timingRoutines import Timer
    from pydsproutines_amd.verifyRoutines import compareValues

    np.random.seed(6)
    timer = Timer()
    length = 1000000
    noise = randnoise(length, 1.0, 1.0, 1.0).astype(np.complex64)
    d_noise = asarray(noise)
    taps = sps.firwin(128, 0.5).astype(np.float32)
    d_taps = asarray(taps)
    timer.start()
    cpu_filt = sps.lfilter(taps, 1.0, noise)
    timer.end("cpu")
    cpkf = CupyKernelFilter()
    timer.start()
    d_filt_smtaps = cpkf.filter_smtaps(d_noise, d_taps)
    timer.end("gpu smtaps")
    timer.start()
    d_filt_smtaps_sminput = cpkf.filter_smtaps_sminput(d_noise, d_taps)
    timer.end("gpu smtaps sminput 128 per blk")
    timer.start()
    d_filt_1024 = cpkf.filter_smtaps_sminput(d_noise, d_taps, OUTPUT_PER_BLK=1024)
    timer.end("gpu smtaps sminput 1028 per blk")
    for d in (d_filt_smtaps, d_filt_smtaps_sminput, d_filt_1024):
        got = d.get()
        assert got.dtype == np.complex64 and got.shape == cpu_filt.shape
        raw, frac = compareValues(cpu_filt, got, verbose=False)
        assert raw <= 2e-5 * np.abs(cpu_filt).max() + 1e-6
    dsr, dsphase = 4, 1
    d_ds = cpkf.filter_smtaps(d_noise, d_taps, dsr=dsr, dsPhase=dsphase)
    raw, _ = compareValues(cpu_filt[dsphase::dsr], d_ds.get(), verbose=False)
    assert raw <= 2e-5 * np.abs(cpu_filt).max() + 1e-6


def test_benchmark_upfirdn_kernels_sequence():
    """benchmark_upfirdnkernels.py:15-67: 100 rows x 1000 samples, firwin(128, 0.2), up from (5, 7, 11), down from (2, 3):
    upfirdn_naive row by row into a preallocated matrix, upfirdn_sm on the whole matrix, scipy.signal.upfirdn per row; the
    script's own assertion fracChg < 1e-4 between the two GPU forms, and the CPU comparison it prints.  Every (up, down)
    pair instead of ten random draws."""
    import scipy.signal as sps

    from pydsproutines_amd import asarray
    from pydsproutines_amd.devarray import zeros
    from pydsproutines_amd.filterRoutines import CupyKernelFilter
    from pydsproutines_amd.signalCreationRoutines import randnoise
    from pydsproutines_amd.verifyRoutines import compareValues

    np.random.seed(7)
    for up in (5, 7, 11):
        for down in (2, 3):
            numRows, length = 100, 1000
            h_x = randnoise(numRows * length, 1, 1, 1).reshape((numRows, length)).astype(np.complex64)
            d_x = asarray(h_x)
            h_taps = sps.firwin(128, 0.2).astype(np.float32)
            d_taps = asarray(h_taps)
            cpkf = CupyKernelFilter()
            d_manualout = zeros((numRows, cpkf.getUpfirdnSize(d_x.shape[1], d_taps.size, up, down)), np.complex64)
            for i in range(numRows):
                cpkf.upfirdn_naive(d_x[i], d_taps, up, down, d_out=d_manualout[i])
            d_out = zeros(d_manualout.shape, np.complex64)
            cpkf.upfirdn_sm(d_x, d_taps, up, down, d_out=d_out)
            h_out = np.vstack([sps.upfirdn(h_taps, h_x[i, :], up, down) for i in range(numRows)])
            assert h_out.shape == d_manualout.shape
            raw, frac = compareValues(h_out.reshape(-1), d_manualout.get().reshape(-1), verbose=False)
            assert raw <= 2e-5 * np.abs(h_out).max() + 1e-6
            rawChg, fracChg = compareValues(d_manualout.get().reshape(-1), d_out.get().reshape(-1), verbose=False)
            assert fracChg < 1e-4  # (the script's own assert, benchmark_upfirdnkernels.py:67)


def test_benchmark_mov_avg_kernels_sequence():
    """benchmark_movAvgKernels.py:19-48: 10^7 float32 samples (mean 2), a 100-tap averaging window as FIR taps through
    filter_smtaps_sminput(THREADS_PER_BLOCK=128, OUTPUT_PER_BLK=256) and as cupyMovingAverage(NUM_PER_THREAD=33,
    THREADS_PER_BLK=32); the script compares the two (compareValues) -- here bounded, and both against NumPy's cumulative sum."""
    from pydsproutines_amd import asarray
    from pydsproutines_amd.filterRoutines import CupyKernelFilter, cupyMovingAverage
    from pydsproutines_amd.verifyRoutines import compareValues

    rng = np.random.default_rng(8)
    length = 10000000
    x = rng.standard_normal(length).astype(np.float32) + np.float32(2.0)
    avgLength = 100
    avgTaps = np.ones(avgLength).astype(np.float32) / avgLength
    d_x, d_avgTaps = asarray(x), asarray(avgTaps)
    cpkf = CupyKernelFilter()
    d_old = cpkf.filter_smtaps_sminput(d_x, d_avgTaps, THREADS_PER_BLOCK=128, OUTPUT_PER_BLK=256)
    d_new = cupyMovingAverage(d_x, avgLength, NUM_PER_THREAD=33, THREADS_PER_BLK=32)
    old, new = d_old.get(), d_new.get()
    assert old.shape == new.shape == (length,) and new.dtype == np.float32
    raw, frac = compareValues(old, new, verbose=False)
    assert raw <= 2e-5
    c = np.concatenate(([0.0], np.cumsum(x.astype(np.float64))))
    ref = (c[1:] - np.concatenate((np.zeros(avgLength), c[1 : length - avgLength + 1]))) / avgLength
    assert np.max(np.abs(new - ref)) <= 2e-5 and np.max(np.abs(old.real - ref)) <= 2e-5
