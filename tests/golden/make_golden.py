#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the importable Python
reference (icyveins7/pydsproutines mounted read-only at /root/reference).

Run ONLY in the build container:

    python tests/golden/make_golden.py

The fixtures are data: seeded inputs plus the reference's outputs for them.
No reference source travels.  The GPU box and the `-m gpu` tests read only the
committed ``*.npz`` files.  While generating, the script also prints how far
the repo's own oracle (``oracle/``) is from the reference on every fixture.
"""

import os
import sys
import contextlib
import io

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PYDSP_REFERENCE", "/root/reference")

os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

with contextlib.redirect_stdout(io.StringIO()):
    import xcorrRoutines as R  # noqa: E402  (the reference)
    import spectralRoutines as RS  # noqa: E402
    import signalCreationRoutines as RC  # noqa: E402

import oracle as O  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def cn(rng, n, dtype=np.complex64):
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2)).astype(dtype)


def qpsk(rng, n, dtype=np.complex64):
    return np.exp(1j * (np.pi / 4 + np.pi / 2 * rng.integers(0, 4, n))).astype(dtype)


def report(name, ref, mine):
    ref = np.asarray(ref)
    mine = np.asarray(mine)
    if ref.dtype.kind in "iu":
        d = int(np.sum(ref != mine))
        print("  %-34s index mismatches: %d / %d" % (name, d, ref.size))
        return d
    err = float(np.max(np.abs(ref - mine))) if ref.size else 0.0
    print("  %-34s max|ref-oracle| = %.3e (max|ref| %.3e)" % (name, err, float(np.max(np.abs(ref))) if ref.size else 0))
    return err


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %s (%.1f KiB)" % (path, os.path.getsize(path) / 1024))


def gen_fastxcorr():
    rng = np.random.default_rng(101)
    n, m = 64, 600
    rx = cn(rng, m)
    cut = (rx[100 : 100 + n] * np.complex64(1.7 - 0.3j)).copy()
    # put a frequency offset of +5 bins on a second copy at delay 300
    rx[300 : 300 + n] += (cut * np.exp(2j * np.pi * 5 * np.arange(n) / n)).astype(np.complex64)
    sub = np.arange(7, 500, 11)
    out = {"cutout": cut, "rx": rx, "shifts_sub": sub}
    print("fastXcorr (N=%d, M=%d)" % (n, m))
    for tag, sh in (("all", None), ("sub", sub)):
        a = quiet(R.fastXcorr, cut, rx, shifts=sh)
        ac = quiet(R.fastXcorr, cut, rx, shifts=sh, absResult=False)
        b, bi = quiet(R.fastXcorr, cut, rx, freqsearch=True, shifts=sh)
        bc, bci = quiet(R.fastXcorr, cut, rx, freqsearch=True, shifts=sh, absResult=False)
        c = quiet(R.fastXcorr, cut, rx, freqsearch=True, outputCAF=True, shifts=sh)
        cc = quiet(R.fastXcorr, cut, rx, freqsearch=True, outputCAF=True, shifts=sh, absResult=False)
        out.update({
            "A_" + tag: a, "Ac_" + tag: ac, "B_" + tag: b, "Bi_" + tag: bi,
            "Bc_" + tag: bc, "Bci_" + tag: bci, "C_" + tag: c, "Cc_" + tag: cc,
        })
        report("A " + tag, a, O.fastXcorr(cut, rx, shifts=sh))
        report("A' " + tag, ac, O.fastXcorr(cut, rx, shifts=sh, absResult=False))
        ob, obi = O.fastXcorr(cut, rx, freqsearch=True, shifts=sh)
        report("B " + tag, b, ob)
        report("B idx " + tag, bi, obi)
        obc, obci = O.fastXcorr(cut, rx, freqsearch=True, shifts=sh, absResult=False)
        report("B' " + tag, bc, obc)
        report("B' idx " + tag, bci, obci)
        report("C " + tag, c, O.fastXcorr(cut, rx, freqsearch=True, outputCAF=True, shifts=sh))
        report("C' " + tag, cc, O.fastXcorr(cut, rx, freqsearch=True, outputCAF=True, shifts=sh, absResult=False))
    # complex128 inputs (the TCC unit test upgrades to 64-bit, xcorrRoutines.py:2180-2184)
    a128 = quiet(R.fastXcorr, cut.astype(np.complex128), rx.astype(np.complex128))
    out["A_all_c128"] = a128
    report("A all (c128)", a128, O.fastXcorr(cut.astype(np.complex128), rx.astype(np.complex128)))
    save("fastxcorr_small", **out)


def gen_kat2():
    # inputs of cython_ext/CyIppXcorrFFT/test_IppXcorrFFT.cpp:9-32 (SURVEY KAT-2)
    i = np.arange(100)
    data = (i + 1j * (i + 1)).astype(np.complex64)
    cut = data[20:50].copy()
    sh = np.arange(0, 70, 3)
    q, fi = quiet(R.fastXcorr, cut, data, freqsearch=True, shifts=sh)
    print("KAT-2")
    oq, ofi = O.fastXcorr(cut, data, freqsearch=True, shifts=sh)
    report("qf2", q, oq)
    report("freqidx", fi, ofi)
    pk, pfi = O.IppXcorrFFT(cut).xcorr(data, 0, 100, 3)
    report("IppXcorrFFT restatement (in range)", q.astype(np.float32), pk[:24])
    assert np.all(pk[24:] == 0) and np.all(pfi[24:] == 0)
    save("kat2_ippxcorrfft", data=data, cutout=cut, shifts=sh, qf2=q, freqidx=fi)


def gen_kat1_kat3():
    # KAT-1: pybinds/ippGroupXcorrCZT/test.py:4,23-32
    data = np.arange(200, dtype=np.float32).view(np.complex64)
    starts = np.array([10, 70])
    lengths = np.array([10, 12])
    g = R.GroupXcorrCZT(data, starts, lengths, -0.1, 0.1, 0.1, 100)
    sh = np.array([9, 10, 11])
    xc, f = g.xcorr(data, sh)
    print("KAT-1")
    og = O.GroupXcorrCZT(data, starts, lengths, -0.1, 0.1, 0.1, 100)
    oxc, of = og.xcorr(data, sh)
    report("GroupXcorrCZT qf2", xc, oxc)
    report("freq", f, of)
    report("ystackNormSq", np.array([g.ystackNormSq]), np.array([og.ystackNormSq]))
    # KAT-3: pybinds/ippCZT/test.py:14-47
    x = (np.arange(10) + 1j * np.arange(10)).astype(np.complex64)
    cz = RS.CZTCached(10, -1, 1, 0.1, 10, convertTo32fc=True)
    y = cz.run(x)
    ocz = O.CZTCached(10, -1, 1, 0.1, 10, convertTo32fc=True)
    print("KAT-3  k=%d nfft=%d" % (cz.k, cz.nfft))
    report("CZTCached.run", y, ocz.run(x))
    assert (ocz.k, ocz.nfft) == (cz.k, cz.nfft)
    save("kat1_kat3_czt", kat1_data=data, kat1_starts=starts, kat1_lengths=lengths, kat1_shifts=sh,
         kat1_qf2=xc, kat1_freq=f, kat1_ynormsq=np.array([g.ystackNormSq]),
         kat3_x=x, kat3_y=y, kat3_k=np.array([cz.k]), kat3_nfft=np.array([cz.nfft]),
         kat3_ww=cz.ww, kat3_fv=cz.fv, kat3_aa=cz.aa)


def gen_czt():
    rng = np.random.default_rng(202)
    n, m = 96, 700
    fs = 1000.0
    rx = cn(rng, m)
    cut = rx[200 : 200 + n].copy()
    rx[200 : 200 + n] *= np.exp(2j * np.pi * 3.3 * np.arange(n) / fs).astype(np.complex64)
    sh = np.arange(150, 260, 3)
    caf, f = quiet(R.cztXcorr, cut, rx, -20.0, 20.0, fs, cztStep=0.5, outputCAF=True, shifts=sh)
    res, fpk = quiet(R.cztXcorr, cut, rx, -20.0, 20.0, fs, cztStep=0.5, outputCAF=False, shifts=sh)
    print("cztXcorr")
    ocaf, of = O.cztXcorr(cut, rx, -20.0, 20.0, fs, cztStep=0.5, outputCAF=True, shifts=sh)
    ores, ofpk = O.cztXcorr(cut, rx, -20.0, 20.0, fs, cztStep=0.5, outputCAF=False, shifts=sh)
    report("CAF", caf, ocaf)
    report("freqs", f, of)
    report("flattened value", res, ores)
    report("flattened freq", fpk, ofpk)
    # stand-alone czt + CZTCached on random data, f64 and 32fc constants
    x = cn(rng, 300, np.complex128)
    y1 = RS.czt(x, -30.0, 30.0, 0.25, fs)
    cz = RS.CZTCached(300, -30.0, 30.0, 0.25, fs)
    y2 = cz.run(x)
    report("czt", y1, O.czt(x, -30.0, 30.0, 0.25, fs))
    report("CZTCached.run f64", y2, O.CZTCached(300, -30.0, 30.0, 0.25, fs).run(x))
    xm = cn(rng, 5 * 300).reshape(5, 300)
    cz32 = RS.CZTCached(300, -30.0, 30.0, 0.25, fs, convertTo32fc=True)
    ym = cz32.runMany(xm)
    report("CZTCached.runMany 32fc", ym, O.CZTCached(300, -30.0, 30.0, 0.25, fs, convertTo32fc=True).runMany(xm))
    save("cztxcorr_small", cutout=cut, rx=rx, shifts=sh, fs=np.array([fs]), caf=caf, freqs=f, res=res, fpk=fpk,
         czt_x=x, czt_y=y1, cached_y=y2, many_x=xm, many_y=ym)


def gen_group():
    rng = np.random.default_rng(303)
    fs = 2048.0
    m = 3000
    y = qpsk(rng, 900)
    starts = np.array([100, 400, 650])
    lengths = np.array([128, 100, 128])
    rx = (0.5 * cn(rng, m)).astype(np.complex64)
    d0 = 777
    tone = np.exp(2j * np.pi * 6.0 * np.arange(900) / fs)
    for s, l in zip(starts, lengths):
        rx[d0 + s - starts[0] : d0 + s - starts[0] + l] += (y[s : s + l] * tone[s - starts[0] : s - starts[0] + l]).astype(np.complex64)
    freqs = np.arange(-16.0, 16.0, 1.0)
    sh = np.arange(700, 860)
    g = R.GroupXcorr(y, starts, lengths, freqs, fs)
    xc, fpk = g.xcorr(rx, sh)
    print("GroupXcorr")
    og = O.GroupXcorr(y, starts, lengths, freqs, fs)
    oxc, ofpk = og.xcorr(rx, sh)
    report("xc", xc, oxc)
    report("freqpeaks", fpk, ofpk)
    # equal-length groups on the makeFreq grid: GroupXcorrFFT == GroupXcorr(freqs=makeFreq)
    L, fftlen = 128, 256
    st2 = np.array([50, 300, 600])
    yg = np.stack([y[s : s + L] for s in st2])
    rx2 = (0.5 * cn(rng, m)).astype(np.complex64)
    for s in st2:
        rx2[500 + s - st2[0] : 500 + s - st2[0] + L] += y[s : s + L]
    sh2 = np.arange(440, 560)
    mf = RC.makeFreq(fftlen, fs)
    g2 = R.GroupXcorr(y, st2, np.array([L, L, L]), mf, fs)
    xc2, fpk2 = g2.xcorr(rx2, sh2)
    of = O.GroupXcorrFFT(yg, st2, fs, fftlen=fftlen)
    oxc2, ofi2 = of.xcorr(rx2, sh2)
    report("GroupXcorrFFT vs GroupXcorr(makeFreq)", xc2, oxc2)
    report("  freq", fpk2, mf[ofi2])
    save("groupxcorr_small", y=y, starts=starts, lengths=lengths, rx=rx, freqs=freqs, fs=np.array([fs]), shifts=sh,
         xc=xc, freqpeaks=fpk, st2=st2, yg=yg, rx2=rx2, sh2=sh2, fftlen=np.array([fftlen]), xc2=xc2, fpk2=fpk2)


def gen_c2_mini():
    """Scaled-down config C2: N=256 template, M=8192 rx, F=32 on-grid bins, 0 dB SNR, planted
    at (d0, k0); golden = the reference's own CAF columns."""
    rng = np.random.default_rng(404)
    n, m, k0, d0 = 256, 8192, 5, 3000
    t = qpsk(rng, n)
    rx = cn(rng, m)
    rx[d0 : d0 + n] += (t * np.exp(2j * np.pi * k0 * np.arange(n) / n)).astype(np.complex64)
    bins = np.arange(-16, 16)
    sh = np.concatenate((np.arange(0, 64), np.arange(d0 - 40, d0 + 40), np.arange(m - n + 1 - 64, m - n + 1)))
    caf = quiet(R.fastXcorr, t, rx, freqsearch=True, outputCAF=True, shifts=sh)[:, np.mod(bins, n)]
    g = R.GroupXcorr(t, np.array([0]), np.array([n]), bins * (float(n) / n), float(n))
    xc, fpk = g.xcorr(rx, sh[:-1])
    print("C2-mini")
    report("caf_bins", caf, O.caf_bins(t, rx, bins, sh))
    report("overlap-save (same algorithm, CPU)", caf, O.caf_overlap_save(t, rx, bins, block=1024)[sh])
    save("c2_mini", template=t, rx=rx, bins=bins, shifts=sh, caf=caf, d0=np.array([d0]), k0=np.array([k0]),
         gx_xc=xc, gx_fpk=fpk)


def gen_c1():
    """Config C1: one 1024-sample template vs 65536-sample rx, no frequency search."""
    rng = np.random.default_rng(0)
    rx = cn(rng, 65536)
    d0 = 1000
    cut = rx[d0 : d0 + 1024].copy()
    q = quiet(R.fastXcorr, cut, rx)
    print("C1")
    report("fastXcorr default", q, O.fastXcorr(cut, rx))
    save("c1_fastxcorr", rx=rx, d0=np.array([d0]), qf2=q.astype(np.float32), argmax=np.array([int(np.argmax(q))]),
         peak=np.array([q.max()]))


def gen_czt_offgrid():
    """cztXcorr on a grid whose span is not a whole number of steps ((f2 - f1) / step = 10.53): CZTCached then
    evaluates at f1 + i (f2 - f1 + step) / k while labelling the bins f1 + i step (spectralRoutines.py:239-311)."""
    rng = np.random.default_rng(909)
    n, m, fs = 64, 600, 100.0
    rx = cn(rng, m)
    cut = rx[100 : 100 + n].copy()
    rx[100 : 100 + n] *= np.exp(2j * np.pi * 1.9 * np.arange(n) / fs).astype(np.complex64)
    sh = np.arange(80, 125)
    f1, f2, step = -3.0, 4.37, 0.7
    caf, f = quiet(R.cztXcorr, cut, rx, f1, f2, fs, cztStep=step, outputCAF=True, shifts=sh)
    res, fpk = quiet(R.cztXcorr, cut, rx, f1, f2, fs, cztStep=step, outputCAF=False, shifts=sh)
    print("cztXcorr, off-grid span")
    ocaf, of = O.cztXcorr(cut, rx, f1, f2, fs, cztStep=step, outputCAF=True, shifts=sh)
    ores, ofpk = O.cztXcorr(cut, rx, f1, f2, fs, cztStep=step, outputCAF=False, shifts=sh)
    report("CAF", caf, ocaf)
    report("freqs", f, of)
    report("flattened value", res, ores)
    report("flattened freq", fpk, ofpk)
    save("cztxcorr_offgrid", cutout=cut, rx=rx, shifts=sh, grid=np.array([f1, f2, step, fs]), caf=caf, freqs=f, res=res,
         fpk=fpk)


def gen_kat4():
    """TemplateCrossCorrelator unit test inputs (xcorrRoutines.py:2130-2241): the class itself
    needs cupy upstream; its own test pins it to sqrt(fastXcorr) and |fastXcorr(absResult=False)|."""
    rng = np.random.default_rng(505)
    x = qpsk(rng, 100)
    t1 = (x[20:40] * np.float32(1.234)).astype(np.complex64)
    t2 = (x[40:60] * np.float32(2.345)).astype(np.complex64)
    chk = np.sqrt(quiet(R.fastXcorr, t1.astype(np.complex128), x.astype(np.complex128)))
    a1 = np.abs(quiet(R.fastXcorr, t1, x, absResult=False))
    a2 = np.abs(quiet(R.fastXcorr, t2, x, absResult=False))
    print("KAT-4")
    tcc = O.TemplateCrossCorrelator(np.vstack((t1, t2)), 100)
    out = tcc.correlate(x)
    report("TCC restatement |row0| vs sqrt(fastXcorr)", chk, np.abs(out[0]))
    report("TCC restatement |row1|", a2, np.abs(out[1]))
    save("kat4_tcc", x=x, t1=t1, t2=t2, qf_single=chk, abs1=a1, abs2=a2)


def gen_perm():
    """GroupXcorrCZT_Permutations / GroupXcorrGPU are defined under `import cupy` upstream and cannot be
    imported here.  They are pinned through identities with importable reference classes:
      getCAF(selection) == GroupXcorrCZT(composite of the selected templates, same CZT grid).xcorr
      GroupXcorrGPU.xcorr / xcorrKernel == GroupXcorr.xcorr (value, frequency) on the same inputs."""
    rng = np.random.default_rng(606)
    fs = 4096.0
    L, m = 96, 1400
    groupStarts = np.array([0, 150])
    ygroupIdxs = np.array([0, 0, 1, 1, 1])
    ygroups = np.stack([qpsk(rng, L) for _ in range(5)])
    rx = (0.4 * cn(rng, m)).astype(np.complex64)
    d0, f0 = 500, 22.0
    tone = np.exp(2j * np.pi * f0 * np.arange(groupStarts[-1] + L) / fs)
    for g, t in ((0, 1), (1, 4)):  # the planted permutation: second template of group 0, third of group 1
        s = groupStarts[g]
        rx[d0 + s : d0 + s + L] += (ygroups[t] * tone[s : s + L]).astype(np.complex64)
    f1, f2, bw = -40.0, 40.0, 2.0
    sh = np.arange(440, 560)
    op = O.GroupXcorrCZT_Permutations(ygroups, ygroupIdxs, groupStarts, f1, f2, bw, fs)
    cztf = op.xcorr(rx, sh)
    sels = np.array([[0, 0], [1, 2], [1, 0]])
    cafs = []
    print("GroupXcorrCZT_Permutations (via reference GroupXcorrCZT on the composite)")
    for sel in sels:
        tsel = [np.argwhere(ygroupIdxs == g)[sel[g]][0] for g in range(2)]
        comp = np.zeros(groupStarts[-1] + L, np.complex64)
        for g, t in enumerate(tsel):
            comp[groupStarts[g] : groupStarts[g] + L] = ygroups[t]
        r = R.GroupXcorrCZT(comp, groupStarts, np.array([L, L]), f1, f2, bw, fs)
        caf, f = r.xcorr(rx, sh)
        cafs.append(caf)
        report("  getCAF %s" % (sel,), caf, op.getCAF(sel))
    # GroupXcorrGPU == GroupXcorr
    freqs = np.arange(-32.0, 32.0, 2.0)
    comp = np.zeros(groupStarts[-1] + L, np.complex64)
    comp[:L], comp[groupStarts[1] : groupStarts[1] + L] = ygroups[1], ygroups[4]
    rg = R.GroupXcorr(comp, groupStarts, np.array([L, L]), freqs, fs)
    xc, fpk = rg.xcorr(rx, sh)
    og = O.GroupXcorrGPU(comp, groupStarts, np.array([L, L]), freqs, fs)
    kxc, kfi = og.xcorrKernel(rx, sh)
    report("  GroupXcorrGPU.xcorrKernel value", xc, kxc)
    report("  GroupXcorrGPU.xcorrKernel freq", fpk, freqs[kfi])
    save("perm_small", ygroups=ygroups, ygroupIdxs=ygroupIdxs, groupStarts=groupStarts, rx=rx, shifts=sh,
         fs=np.array([fs]), f1f2bw=np.array([f1, f2, bw]), sels=sels, cafs=np.stack(cafs), cztFreq=cztf,
         gfreqs=freqs, gcomp=comp, gxc=xc, gfpk=fpk, d0f0=np.array([d0, f0]))


def gen_finesearch():
    """fineFreqTimeSearch / GenXcorr (xcorrRoutines.py:583-719): a band-limited signal delayed by a fraction
    of a sample and offset in frequency; two levels of fine frequency grids, a 201-point time scan."""
    rng = np.random.default_rng(707)
    fs, n = 1.0e4, 2048
    spec = np.zeros(n, np.complex128)
    band = np.abs(RC.makeFreq(n, fs)) < 0.3 * fs
    spec[band] = (rng.standard_normal(band.sum()) + 1j * rng.standard_normal(band.sum()))
    x = np.fft.ifft(spec)
    tau, df = 0.37 / fs, 1.7  # y(t) = x(t - tau) e^{j 2 pi df t}
    y = np.fft.ifft(spec * np.exp(-2j * np.pi * RC.makeFreq(n, fs) * tau)) * np.exp(2j * np.pi * df * np.arange(n) / fs)
    x = (x / np.abs(x).max() + 0.01 * cn(rng, n, np.complex128)).astype(np.complex64)
    y = (y / np.abs(y).max() + 0.01 * cn(rng, n, np.complex128)).astype(np.complex64)
    td = np.arange(-1.0, 1.0 + 1e-9, 0.01) / fs
    ff, tdiff, cost = quiet(R.fineFreqTimeSearch, x, y, [0.5, 0.05], 0.0, 4.0, fs, td)
    off, otd, ocost = O.fineFreqTimeSearch(x, y, [0.5, 0.05], 0.0, 4.0, fs, td)
    print("fineFreqTimeSearch: finefreq %r timediff %r (samples %.3f)" % (ff, tdiff, tdiff * fs))
    report("  finefreq", np.array([ff]), np.array([off]))
    report("  timediff", np.array([tdiff]), np.array([otd]))
    report("  cost_vec", cost, ocost)
    ff0, tdiff0, cost0 = quiet(R.fineFreqTimeSearch, x, y, [], 0.0, 4.0, fs, td, None, np.array([-2000.0, 2000.0]))
    _, otd0, ocost0 = O.fineFreqTimeSearch(x, y, [], 0.0, 4.0, fs, td, None, np.array([-2000.0, 2000.0]))
    report("  (no fine freq, bounded) cost_vec", cost0, ocost0)
    gx = R.GenXcorr(td, fs, n)
    gtd, gcost = gx.xcorr(x, y)
    ogx = O.GenXcorr(td, fs, n)
    ogtd, ogcost = ogx.xcorr(x, y)
    report("  GenXcorr cost_vec", gcost, ogcost)
    assert ff0 is None and otd0 == tdiff0 and ogtd == gtd and otd == tdiff and off == ff
    save("finesearch", x=x, y=y, td=td, fs=np.array([fs]), fineRes=np.array([0.5, 0.05]), freqRes=np.array([4.0]),
         finefreq=np.array([ff]), timediff=np.array([tdiff]), cost=cost, timediff0=np.array([tdiff0]), cost0=cost0,
         bounds=np.array([-2000.0, 2000.0]), gen_timediff=np.array([gtd]), gen_cost=gcost)


def gen_dottones():
    """dotTonesScaling_32f (cupy-only upstream): pinned by the identity of its own docstring,
    sum over blocks == czt(src, f1, f2, fstep, fs) with f0 = -f1/fs, step = -fstep/fs."""
    rng = np.random.default_rng(808)
    fs, n = 1000.0, 1000  # 15 full blocks + one of 40 samples
    src = cn(rng, n)
    f1, f2, fstep = -20.0, 19.5, 0.5
    ref = quiet(RS.czt, src.astype(np.complex128), f1, f2, fstep, fs)
    k = ref.size
    mine = O.kernels.dotTonesScaling(-f1 / fs, -fstep / fs, k, src).sum(axis=0)
    print("dotTonesScaling (block sums vs reference czt), k=%d" % k)
    report("  sum over blocks", ref, mine)
    save("dottones", src=src, fs=np.array([fs]), f1f2step=np.array([f1, f2, fstep]), czt=ref)


if __name__ == "__main__":
    gen_dottones()
    gen_finesearch()
    gen_perm()
    gen_fastxcorr()
    gen_kat2()
    gen_kat1_kat3()
    gen_czt()
    gen_czt_offgrid()
    gen_group()
    gen_c2_mini()
    gen_c1()
    gen_kat4()
