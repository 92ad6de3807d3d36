"""CPU tests: the oracle (oracle/) against the committed golden vectors, which were
produced by the importable Python reference (tests/golden/make_golden.py), and against
the known-answer values of SURVEY.md Appendix A."""

import numpy as np
import pytest

import oracle as O

TOL = 2e-6  # oracle accumulates energies in f64, the reference in f32 -> ~1e-7 relative


@pytest.mark.parametrize("tag", ["all", "sub"])
def test_fastxcorr_six_branches(golden, tag):
    g = golden("fastxcorr_small")
    cut, rx = g["cutout"], g["rx"]
    sh = None if tag == "all" else g["shifts_sub"]
    np.testing.assert_allclose(O.fastXcorr(cut, rx, shifts=sh), g["A_" + tag], atol=TOL)
    np.testing.assert_allclose(O.fastXcorr(cut, rx, shifts=sh, absResult=False), g["Ac_" + tag], atol=TOL)
    b, bi = O.fastXcorr(cut, rx, freqsearch=True, shifts=sh)
    assert b.dtype == np.float64 and bi.dtype == np.uint32
    np.testing.assert_allclose(b, g["B_" + tag], atol=TOL)
    np.testing.assert_array_equal(bi, g["Bi_" + tag])
    bc, bci = O.fastXcorr(cut, rx, freqsearch=True, shifts=sh, absResult=False)
    assert bc.dtype == np.complex128
    np.testing.assert_allclose(bc, g["Bc_" + tag], atol=TOL)
    np.testing.assert_array_equal(bci, g["Bci_" + tag])
    np.testing.assert_allclose(O.fastXcorr(cut, rx, True, True, sh), g["C_" + tag], atol=TOL)
    np.testing.assert_allclose(O.fastXcorr(cut, rx, True, True, sh, False), g["Cc_" + tag], atol=TOL)


def test_fastxcorr_complex128(golden):
    g = golden("fastxcorr_small")
    out = O.fastXcorr(g["cutout"].astype(np.complex128), g["rx"].astype(np.complex128))
    np.testing.assert_allclose(out, g["A_all_c128"], atol=1e-14)


def test_branch_b_is_rowmax_of_branch_c(golden):
    g = golden("fastxcorr_small")
    c = O.fastXcorr(g["cutout"], g["rx"], True, True)
    b, bi = O.fastXcorr(g["cutout"], g["rx"], True, False)
    np.testing.assert_array_equal(bi, np.argmax(c, axis=1))
    np.testing.assert_allclose(b, c.max(axis=1), rtol=1e-12)


def test_kat2_ippxcorrfft(golden):
    g = golden("kat2_ippxcorrfft")
    # SURVEY Appendix A KAT-2 literal values (first / best / last)
    np.testing.assert_allclose(g["qf2"][[0, 7, 23]], [0.923214853, 0.999957860, 0.980599701], atol=2e-7)
    assert np.all(g["freqidx"] == 0)
    pk, fi = O.IppXcorrFFT(g["cutout"], num_threads=3).xcorr(g["data"], 0, 100, 3)
    assert pk.dtype == np.float32 and fi.dtype == np.int32 and pk.size == 34
    np.testing.assert_allclose(pk[:24], g["qf2"], atol=TOL)
    np.testing.assert_array_equal(fi[:24], g["freqidx"])
    # out-of-range delays report (0, 0) instead of raising (IppXcorrFFT.cpp:125-130)
    assert np.all(pk[24:] == 0) and np.all(fi[24:] == 0)


def test_kat1_groupxcorrczt(golden):
    g = golden("kat1_kat3_czt")
    kat1 = np.array([[0.995349501, 0.999908180, 0.995299135],
                     [0.995195581, 0.999999890, 0.995195581],
                     [0.994872691, 0.999913669, 0.994921216]])
    np.testing.assert_allclose(g["kat1_qf2"], kat1, atol=5e-7)
    obj = O.GroupXcorrCZT(g["kat1_data"], g["kat1_starts"], g["kat1_lengths"], -0.1, 0.1, 0.1, 100)
    xc, f = obj.xcorr(g["kat1_data"], g["kat1_shifts"])
    np.testing.assert_allclose(xc, g["kat1_qf2"], atol=TOL)
    np.testing.assert_allclose(f, [-0.1, 0.0, 0.1], atol=1e-12)
    assert abs(obj.ystackNormSq - 570074.06) < 0.5


def test_kat3_cztcached(golden):
    g = golden("kat1_kat3_czt")
    cz = O.CZTCached(10, -1, 1, 0.1, 10, convertTo32fc=True)
    assert (cz.k, cz.nfft) == (21, 30) == (int(g["kat3_k"][0]), int(g["kat3_nfft"][0]))
    y = cz.run(g["kat3_x"])
    np.testing.assert_allclose(y, g["kat3_y"], atol=1e-5)
    np.testing.assert_allclose(y[:2], [10.388418 - 20.38842j, -1.9021038 - 27.646591j], atol=2e-5)
    assert int(np.argmax(np.abs(y))) == 10 and abs(np.abs(y).max() - 63.63962) < 1e-4
    for name in ("ww", "fv", "aa"):
        np.testing.assert_allclose(getattr(cz, name), g["kat3_" + name], atol=1e-6)


def test_cztxcorr_offgrid_span(golden):
    """(f2 - f1) / step = 10.53: CZTCached's chirp rate stretches the evaluated grid, the labels stay f1 + i step."""
    g = golden("cztxcorr_offgrid")
    f1, f2, step, fs = (float(v) for v in g["grid"])
    caf, f = O.cztXcorr(g["cutout"], g["rx"], f1, f2, fs, step, True, g["shifts"])
    assert caf.shape == (45, 11)
    np.testing.assert_allclose(caf, g["caf"], atol=TOL)
    np.testing.assert_array_equal(f, g["freqs"])
    res, fpk = O.cztXcorr(g["cutout"], g["rx"], f1, f2, fs, step, False, g["shifts"])
    np.testing.assert_allclose(res, g["res"], atol=TOL)
    np.testing.assert_array_equal(fpk, g["fpk"])


def test_cztxcorr_and_czt(golden):
    g = golden("cztxcorr_small")
    fs = float(g["fs"][0])
    caf, f = O.cztXcorr(g["cutout"], g["rx"], -20.0, 20.0, fs, 0.5, True, g["shifts"])
    np.testing.assert_allclose(caf, g["caf"], atol=TOL)
    np.testing.assert_array_equal(f, g["freqs"])
    res, fpk = O.cztXcorr(g["cutout"], g["rx"], -20.0, 20.0, fs, 0.5, False, g["shifts"])
    assert res.dtype == np.complex64
    np.testing.assert_allclose(res, g["res"], atol=TOL)
    np.testing.assert_array_equal(fpk, g["fpk"])
    np.testing.assert_allclose(O.czt(g["czt_x"], -30.0, 30.0, 0.25, fs), g["czt_y"], atol=1e-10)
    np.testing.assert_allclose(O.CZTCached(300, -30.0, 30.0, 0.25, fs).run(g["czt_x"]), g["cached_y"], atol=1e-10)
    ym = O.CZTCached(300, -30.0, 30.0, 0.25, fs, convertTo32fc=True).runMany(g["many_x"])
    np.testing.assert_allclose(ym, g["many_y"], atol=1e-4)
    # CZT against the brute-force DFT (the reference's tests/compare_czt_impl.py check)
    cz = O.CZTCached(300, -30.0, 30.0, 0.25, fs)
    np.testing.assert_allclose(cz.run(g["czt_x"]), O.dft(g["czt_x"], cz.getFreq(), fs), atol=1e-9)


def test_groupxcorr_and_fft_equivalence(golden):
    g = golden("groupxcorr_small")
    fs = float(g["fs"][0])
    obj = O.GroupXcorr(g["y"], g["starts"], g["lengths"], g["freqs"], fs)
    xc, fpk = obj.xcorr(g["rx"], g["shifts"])
    np.testing.assert_allclose(xc, g["xc"], atol=TOL)
    np.testing.assert_array_equal(fpk, g["freqpeaks"])
    assert fpk[np.argmax(xc)] == 6.0 and g["shifts"][np.argmax(xc)] == 777
    # GroupXcorrFFT == GroupXcorr(freqs=makeFreq(fftlen, fs))  
    fftlen = int(g["fftlen"][0])
    of = O.GroupXcorrFFT(g["yg"], g["st2"], fs, fftlen=fftlen)
    xc2, fi2 = of.xcorr(g["rx2"], g["sh2"])
    np.testing.assert_allclose(xc2, g["xc2"], atol=TOL)
    np.testing.assert_array_equal(O.makeFreq(fftlen, fs)[fi2], g["fpk2"])
    full = of.xcorr(g["rx2"], g["sh2"], flattenToTime=False)
    np.testing.assert_allclose(full.max(axis=1), xc2, rtol=1e-12)
    # native twin: float32 full plane
    nat = O.IppGroupXcorrFFT(g["yg"], g["st2"].astype(np.int32), int(fs), fftlen).xcorr(g["rx2"], g["sh2"])
    assert nat.dtype == np.float32 and nat.shape == (g["sh2"].size, fftlen)
    np.testing.assert_allclose(nat, full, atol=5e-6)
    with pytest.raises(ValueError):
        O.IppGroupXcorrFFT(g["yg"], g["st2"], int(fs), fftlen=64)


def test_c2_mini_surface_and_overlap_save(golden):
    g = golden("c2_mini")
    t, rx, bins, sh = g["template"], g["rx"], g["bins"], g["shifts"]
    caf = O.caf_bins(t, rx, bins, sh)
    np.testing.assert_allclose(caf, g["caf"], atol=TOL)
    d0, k0 = int(g["d0"][0]), int(g["k0"][0])
    r, c = np.unravel_index(np.argmax(caf), caf.shape)
    assert (sh[r], bins[c]) == (d0, k0)
    # hypothesis-domain overlap-save (what the GPU computes) == per-delay FFT columns
    os_ = O.caf_overlap_save(t, rx, bins, block=1024)
    np.testing.assert_allclose(os_[sh], g["caf"], atol=1e-6)
    # GroupXcorr on the same grid gives the row maxima and the peak frequency in Hz
    gx = O.GroupXcorr(t, np.array([0]), np.array([t.size]), bins.astype(float), float(t.size))
    xc, fpk = gx.xcorr(rx, sh[:-1])
    np.testing.assert_allclose(xc, g["gx_xc"], atol=TOL)
    np.testing.assert_array_equal(fpk, g["gx_fpk"])


def test_c1(golden):
    g = golden("c1_fastxcorr")
    d0 = int(g["d0"][0])
    rx = g["rx"]
    q = O.fastXcorr(rx[d0 : d0 + 1024].copy(), rx)
    assert q.shape == (64513,) and q.dtype == np.float64
    np.testing.assert_allclose(q, g["qf2"], atol=1e-6)
    assert int(np.argmax(q)) == d0 == int(g["argmax"][0]) and abs(q[d0] - 1.0) < 1e-6


def test_kat4_tcc(golden):
    g = golden("kat4_tcc")
    x, t1, t2 = g["x"], g["t1"], g["t2"]
    with pytest.raises(ValueError):
        O.TemplateCrossCorrelator(t1, 100)
    one = O.TemplateCrossCorrelator(t1.reshape(1, -1), 100)
    qf, ti = one.correlate(x, returnMax=True)
    assert qf.size == 81 and np.all(ti == 0)
    np.testing.assert_array_almost_equal(qf, g["qf_single"], decimal=6)
    both = O.TemplateCrossCorrelator(np.vstack((t1, t2)), 100)
    out = both.correlate(x)
    np.testing.assert_array_almost_equal(np.abs(out[0]), g["abs1"], decimal=6)
    np.testing.assert_array_almost_equal(np.abs(out[1]), g["abs2"], decimal=6)
    qf, ti = both.correlate(x, returnMax=True)
    np.testing.assert_array_equal(qf, np.abs(out).max(axis=0))
    np.testing.assert_array_equal(ti, np.argmax(np.abs(out), axis=0))
    assert abs(qf[20] - 1) < 1e-5 and ti[20] == 0 and abs(qf[40] - 1) < 1e-5 and ti[40] == 1
    with pytest.raises(ValueError):
        both.correlate(x[:50])


def test_permutations_and_groupxcorrgpu(golden):
    """GroupXcorrCZT_Permutations / GroupXcorrGPU (cupy-only upstream): pinned through the reference's
    GroupXcorrCZT on the composite of the selected templates and GroupXcorr on the same inputs."""
    g = golden("perm_small")
    f1, f2, bw = g["f1f2bw"]
    fs = float(g["fs"][0])
    op = O.GroupXcorrCZT_Permutations(g["ygroups"], g["ygroupIdxs"], g["groupStarts"], f1, f2, bw, fs)
    f = op.xcorr(g["rx"], g["shifts"])
    np.testing.assert_allclose(f, g["cztFreq"])
    assert op.xcTemplates.shape == (5, g["shifts"].size, f.size) and op.rxgroupNormSq.shape == (2, g["shifts"].size)
    for sel, caf in zip(g["sels"], g["cafs"]):
        np.testing.assert_allclose(op.getCAF(sel), caf, atol=5e-7)
    d0, f0 = g["d0f0"]
    i, j = np.unravel_index(np.argmax(g["cafs"][1]), g["cafs"][1].shape)  # the planted permutation [1, 2]
    assert g["shifts"][i] == d0 and f[j] == f0 and g["cafs"][1].max() > 3 * g["cafs"][0].max()
    with pytest.raises(AssertionError):
        op.getCAF(np.array([0]))
    with pytest.raises(AssertionError):
        O.GroupXcorrCZT_Permutations(g["ygroups"], np.array([0, 0, 2, 2, 2]), g["groupStarts"], f1, f2, bw, fs)
    og = O.GroupXcorrGPU(g["gcomp"], g["groupStarts"], np.array([96, 96]), g["gfreqs"], fs)
    xc, fpk = og.xcorr(g["rx"], g["shifts"])
    np.testing.assert_allclose(xc, g["gxc"], atol=5e-7)
    np.testing.assert_array_equal(fpk, g["gfpk"])
    kxc, kfi = og.xcorrKernel(g["rx"], g["shifts"])
    assert kxc.dtype == np.float32 and kfi.dtype == np.int32
    np.testing.assert_allclose(kxc, g["gxc"], atol=5e-7)
    np.testing.assert_array_equal(g["gfreqs"][kfi], g["gfpk"])
    with pytest.raises(AssertionError):
        og.xcorrKernel(g["rx"], g["shifts"][:-1])  # 119 shifts, 2 per block


def test_finefreqtimesearch_and_genxcorr(golden):
    g = golden("finesearch")
    fs = float(g["fs"][0])
    ff, td, cost = O.fineFreqTimeSearch(g["x"], g["y"], list(g["fineRes"]), 0.0, float(g["freqRes"][0]), fs, g["td"])
    assert ff == g["finefreq"][0] and td == g["timediff"][0]
    np.testing.assert_allclose(cost, g["cost"], atol=1e-12)
    ff0, td0, cost0 = O.fineFreqTimeSearch(g["x"], g["y"], [], 0.0, 4.0, fs, g["td"], None, g["bounds"])
    assert ff0 is None and td0 == g["timediff0"][0]
    np.testing.assert_allclose(cost0, g["cost0"], atol=1e-12)
    gx = O.GenXcorr(g["td"], fs, g["x"].size)
    gtd, gcost = gx.xcorr(g["x"], g["y"])
    assert gtd == g["gen_timediff"][0]
    np.testing.assert_allclose(gcost, g["gen_cost"], atol=1e-12)
    # an explicit steering matrix is the same as the default one
    sv = O.makeTimeScanSteervec(g["td"], fs, g["x"].size)
    _, td1, cost1 = O.fineFreqTimeSearch(g["x"], g["y"], [], 0.0, 4.0, fs, g["td"], sv)
    np.testing.assert_allclose(cost1, gcost, atol=1e-12)
    assert td1 == gtd


def test_dottones_block_sums_are_the_czt(golden):
    g = golden("dottones")
    f1, f2, step = g["f1f2step"]
    fs = float(g["fs"][0])
    out = O.kernels.dotTonesScaling(-f1 / fs, -step / fs, g["czt"].size, g["src"])
    assert out.shape == (16, 80)
    np.testing.assert_allclose(out.sum(axis=0), g["czt"], atol=1e-9)
    # a block row is the dot product of its 64 samples only
    i = np.arange(64, 128)
    want = np.sum(g["src"][64:128].astype(np.complex128) * np.exp(2j * np.pi * (-(f1 + 3 * step) / fs) * i))
    assert abs(out[1, 3] - want) < 1e-9


def test_kernel_semantics_against_scipy():
    """The reference pins its kernels to scipy (filterRoutines.py:1256,1319,1358;
    benchmark_upfirdnkernels.py:58-67, benchmark_filterkernels.py:72-74)."""
    import scipy.signal as sps
    from oracle import kernels as K

    rng = np.random.default_rng(7)
    x = rng.standard_normal(5000).astype(np.float32)
    for L in (1, 7, 100, 333):
        np.testing.assert_allclose(K.movingAverage(x, L), sps.lfilter(np.ones(L) / L, 1, x), atol=2e-6)
        np.testing.assert_allclose(K.movingAverage(x, L, True), sps.lfilter(np.ones(L), 1, x), atol=1e-4)
    z = (rng.standard_normal(3000) + 1j * rng.standard_normal(3000)).astype(np.complex64)
    np.testing.assert_allclose(K.movingComplexSum(z, 50), np.abs(np.convolve(z, np.ones(50), "valid")) ** 2, rtol=1e-4)
    taps = sps.firwin(64, 0.2).astype(np.float32)
    np.testing.assert_allclose(K.filter_lfilter(z, taps), sps.lfilter(taps, 1, z), atol=1e-5)
    np.testing.assert_allclose(K.filter_lfilter(z[1000:], taps, delay=z[:1000]), sps.lfilter(taps, 1, z)[1000:], atol=1e-5)
    np.testing.assert_allclose(K.filter_lfilter(z, taps, dsr=4, dsPhase=1), sps.lfilter(taps, 1, z)[1::4], atol=1e-5)
    y = K.upfirdn(z, taps, 3, 2)
    assert y.size == K.upfirdn_size(z.size, taps.size, 3, 2)
    # sliding product rows: unit-norm rows times ||x||
    cut = z[100:164]
    rows = K.slidingMultiplyNormalised(cut.conj(), z, 90, 20)
    ref = np.array([z[s : s + 64] * cut.conj() / np.linalg.norm(z[s : s + 64]) / np.linalg.norm(cut) for s in range(90, 110)])
    np.testing.assert_allclose(rows, ref, atol=1e-6)
    assert abs(abs(rows[10].sum()) - 1.0) < 1e-5
    pk = K.findLocalMaxima(np.array([0, 1, 0, 0.2, 0.1, 3, 2, 5], np.float32), 0.15)
    np.testing.assert_array_equal(pk, [1, 3, 5, 7])
    np.testing.assert_array_equal(K.topk_peaks(np.array([0, 1, 0, 1, 0, 3, 2, 5], np.float32), 0.5, 3), [7, 5, 1])
    am, mx = K.argmaxAbsRows(np.array([[1, 2j, -2, 0], [0, 0, 0, 0]], np.complex64))
    np.testing.assert_array_equal(am, [1, 0])
    np.testing.assert_array_equal(mx, [2, 0])
