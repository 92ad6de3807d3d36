"""The reference's benchmark scripts as call SEQUENCES (layer L4): the same functions, in the same order,
with the scripts' own arguments and cross-checks (verifyRoutines.compareValues between implementations), at sizes
that run in seconds.  Written from the cited call lines, not from the scripts' text; `cp.asarray` becomes this package's
`asarray` (there is no cupy on the box -- the one difference a user of the reference has to make).

    benchmarks/benchmark_xcorrs.py:28-59                      fastXcorr <-> CyIppXcorrFFT <-> cp_fastXcorr
    benchmarks/benchmark_groupXcorrs.py:37-72                 GroupXcorrCZT <-> pbIppGroupXcorrCZT (1 thread, 4 threads)
    benchmarks/benchmark_cupyTemplateCrossCorrelator.py:32-37 TemplateCrossCorrelator(...).correlate(dx, returnMax=True)
    benchmarks/benchmark_czts.py:31-82                        CZTCachedGPU / CZTCached / pbIppCZT32fc / dot-tones kernel
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
