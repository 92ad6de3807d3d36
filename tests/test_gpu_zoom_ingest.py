"""GPU tests: config-5 pipeline (coarse CAF -> top-k local maxima -> CZT fine zoom) against the oracle's
cztXcorr, and the int16 IQ ingest kernel (bit-exact)."""

import os

import numpy as np
import pytest

import oracle as O
from oracle import kernels as K
from conftest import REPO, cn, qpsk

pytestmark = pytest.mark.gpu


def test_c5_caf_plus_czt_zoom():
    from pydsproutines_amd import CAFPlan, asarray
    from pydsproutines_amd.zoom import caf_with_zoom, czt_zoom, topk_local_maxima

    rng = np.random.default_rng(55)
    n, m, fs = 1024, 200_000, 1024.0  # bins are 1 Hz wide
    t = qpsk(rng, n)
    rx = (0.5 * cn(rng, m)).astype(np.complex64)
    planted = [(20_000, 3.30, 1.0), (77_777, -7.65, 0.8), (150_001, 11.02, 0.6), (180_500, 0.48, 0.4)]
    for d, f, a in planted:
        rx[d : d + n] += (a * t * np.exp(2j * np.pi * f * np.arange(n) / fs)).astype(np.complex64)
    bins = np.arange(-16, 16)
    d_rx = asarray(rx)
    plan = CAFPlan(t, max_rx_len=m, bins=bins, grid=n)
    res = plan.run(d_rx, rows=True, peak=True)
    out = caf_with_zoom(plan, d_rx, res, bins, n, fs, k=4, span_bins=1.0, step_bins=1.0 / 64)
    assert sorted(o["delay"] for o in out) == [d for d, _, _ in planted]
    cq = [o["coarse_qf2"] for o in out]
    assert cq == sorted(cq, reverse=True)  # best first (coarse value descending)
    truth = {d: (f, a) for d, f, a in planted}
    for o in out:
        d = o["delay"]
        f, a = truth[d]
        assert abs(o["coarse_bin"] - f) <= 0.51
        assert abs(o["fine_freq"] - f) <= 1.0 / 64 + 1e-9
        assert o["fine_qf2"] >= o["coarse_qf2"] - 1e-6  # the fine grid contains (or brackets) the coarse bin
        # oracle: cztXcorr on the same grid at that delay
        ref, fr = O.cztXcorr(t, rx, o["coarse_bin"] - 1.0, o["coarse_bin"] + 1.0, fs, cztStep=1.0 / 64, outputCAF=True,
                             shifts=np.array([d]))
        assert abs(ref[0].max() - o["fine_qf2"]) <= 1e-4
        assert abs(fr[int(np.argmax(ref[0]))] - o["fine_freq"]) <= 1e-9
    # the building blocks on their own
    idx, vals = topk_local_maxima(res.row_max[0], 3, 0.05)
    oi = K.topk_peaks(res.row_max[0].get(), 0.05, 3)
    np.testing.assert_array_equal(idx, oi)
    ff, fq, planes = czt_zoom(t, d_rx, [planted[1][0]], [-8.0], fs, 1.0, 1.0 / 32)
    ref, fr = O.cztXcorr(t, rx, -9.0, -7.0, fs, cztStep=1.0 / 32, outputCAF=True, shifts=np.array([planted[1][0]]))
    np.testing.assert_allclose(planes[0], ref[0], atol=1e-4)
    with pytest.raises(ValueError):
        topk_local_maxima(res.row_max[0], 3, 0.0, maxNumPeaks=100)  # too many candidates for the buffer


def test_zoom_czt_through_ctypes_only():
    """caf_zoom_czt bound the way INTEGRATION.md shows a reference-side maintainer would bind it: ctypes + NumPy and
    nothing else of this package -- plan create -> execute (rows) -> zoom -> table read-back; checked against the
    oracle's cztXcorr at every returned delay, and against the Python chain's ordering rule (value descending, delay
    ascending)."""
    import ctypes as ct

    # (a CDLL object of its own: the package's loader has typed the functions of the one it holds)
    lib = ct.CDLL(os.path.join(REPO, "pydsproutines_amd", "libcaf.so"))
    P, I32, I64 = ct.c_void_p, ct.c_int32, ct.c_int64

    class Desc(ct.Structure):
        _fields_ = [("num_templates", I32), ("template_len", I32), ("h_templates", P), ("auto_conj", I32), ("num_groups", I32),
                    ("h_group_start", P), ("h_group_len", P), ("freq_mode", I32), ("num_freqs", I32), ("h_bins", P),
                    ("grid", I32), ("h_freqs_norm", P), ("max_rx_len", I64), ("log2_block", I32), ("blocks_per_batch", I32),
                    ("engine", I32), ("reserved", I32)]

    class Outs(ct.Structure):
        _fields_ = [(n, P) for n in ("d_surface", "d_row_max", "d_row_arg", "d_peak_val", "d_peak_delay", "d_peak_freq", "d_cqf")]

    class ZoomOuts(ct.Structure):
        _fields_ = [(n, P) for n in ("d_count", "d_delay", "d_coarse_freq_index", "d_coarse_qf2", "d_fine_index",
                                     "d_fine_freq", "d_fine_qf2", "d_planes")]

    for f in ("caf_plan_create", "caf_plan_execute", "caf_plan_destroy", "caf_malloc", "caf_free", "caf_h2d", "caf_d2h",
              "caf_stream_sync", "caf_zoom_czt", "caf_zoom_num_bins"):
        getattr(lib, f).restype = I32

    def ok(rc):
        assert rc == 0, rc

    def dmalloc(nbytes):
        p = P()
        ok(lib.caf_malloc(ct.byref(p), I64(nbytes)))
        return p

    rng = np.random.default_rng(77)
    n, m, grid = 512, 60_000, 512
    t = qpsk(rng, n)
    rx = (0.5 * cn(rng, m)).astype(np.complex64)
    planted = [(5_000, 2.25, 1.0), (30_123, -5.5, 0.7), (51_000, 6.75, 0.5)]  # (delay, frequency in bins, amplitude)
    for d, f, a in planted:
        rx[d : d + n] += (a * t * np.exp(2j * np.pi * f * np.arange(n) / grid)).astype(np.complex64)
    bins = np.arange(-8, 8, dtype=np.int32)
    S = m - n + 1
    desc = Desc(1, n, t.ctypes.data, 1, 0, None, None, 0, bins.size, bins.ctypes.data, grid, None, m, 0, 0, 0, 0)
    plan = P()
    ok(lib.caf_plan_create(ct.byref(plan), ct.byref(desc)))
    d_rx, d_rmax, d_rarg = dmalloc(m * 8), dmalloc(S * 4), dmalloc(S * 4)
    ok(lib.caf_h2d(d_rx, P(rx.ctypes.data), I64(m * 8), None))
    outs = Outs(None, d_rmax, d_rarg, None, None, None, None)
    ok(lib.caf_plan_execute(plan, d_rx, I64(m), I64(0), I64(S), ct.byref(outs), None))
    k, span, step = 5, 1.0 / grid, 1.0 / 32 / grid
    nb = I32()
    ok(lib.caf_zoom_num_bins(ct.c_double(span), ct.c_double(step), ct.byref(nb)))
    assert nb.value == 65
    bufs = {name: dmalloc(sz) for name, sz in (("cnt", 4), ("dly", 4 * k), ("ci", 4 * k), ("cq", 4 * k), ("fi", 4 * k),
                                                 ("ff", 8 * k), ("fq", 4 * k), ("pl", 4 * k * nb.value))}
    zo = ZoomOuts(bufs["cnt"], bufs["dly"], bufs["ci"], bufs["cq"], bufs["fi"], bufs["ff"], bufs["fq"], bufs["pl"])
    ok(lib.caf_zoom_czt(plan, I32(0), d_rx, I64(m), d_rmax, d_rarg, I64(0), I64(S), I32(k), ct.c_float(0.03), ct.c_double(span),
                        ct.c_double(step), ct.byref(zo), None))
    ok(lib.caf_stream_sync(None))

    def back(name, dtype, count):
        a = np.empty(count, dtype)
        ok(lib.caf_d2h(P(a.ctypes.data), bufs[name], I64(a.nbytes), None))
        return a

    cnt = int(back("cnt", np.int32, 1)[0])
    dly, ci, cq = back("dly", np.int32, k), back("ci", np.int32, k), back("cq", np.float32, k)
    fi, ff, fq = back("fi", np.int32, k), back("ff", np.float64, k), back("fq", np.float32, k)
    pl = back("pl", np.float32, k * nb.value).reshape(k, nb.value)
    rmax = np.empty(S, np.float32)
    ok(lib.caf_d2h(P(rmax.ctypes.data), d_rmax, I64(S * 4), None))
    assert cnt == 3 and list(dly[cnt:]) == [-1, -1]
    assert sorted(dly[:cnt]) == [d for d, _, _ in planted]
    assert list(cq[:cnt]) == sorted(cq[:cnt], reverse=True)          # best first
    np.testing.assert_array_equal(dly[:cnt], K.topk_peaks(rmax, 0.03, k))  # the oracle's ordering rule
    for i in range(cnt):
        d = int(dly[i])
        f_true = dict((dd, f) for dd, f, _ in planted)[d]
        cb = int(bins[ci[i]])
        assert cq[i] == rmax[d] and abs(cb - f_true) <= 0.51
        ref, fr = O.cztXcorr(t, rx, cb - 1.0, cb + 1.0, float(grid), cztStep=1.0 / 32, outputCAF=True, shifts=np.array([d]))
        np.testing.assert_allclose(pl[i], ref[0], atol=1e-4)
        assert fi[i] == int(np.argmax(pl[i])) and fq[i] == pl[i].max()
        assert abs(ff[i] * grid - fr[fi[i]]) <= 1e-9 and abs(ff[i] * grid - f_true) <= 1.0 / 32 + 1e-9
    for b in list(bufs.values()) + [d_rx, d_rmax, d_rarg]:
        ok(lib.caf_free(b))
    ok(lib.caf_plan_destroy(plan))


def test_iq16_ingest(tmp_path):
    from pydsproutines_amd import asarray
    from pydsproutines_amd.usrpRoutines import iq16_to_complex64, simpleBinRead, simpleBinReadToDevice

    rng = np.random.default_rng(66)
    for n in (0, 1, 3, 4, 1023, 100_003):
        raw = rng.integers(-32768, 32768, 2 * n, dtype=np.int16)
        got = iq16_to_complex64(asarray(raw)).get()
        np.testing.assert_array_equal(got, raw.astype(np.float32).view(np.complex64))  # bit-exact
    raw = rng.integers(-2048, 2048, 2 * 5000, dtype=np.int16)
    np.testing.assert_array_equal(iq16_to_complex64(asarray(raw), 1.0 / 2048).get(),
                                  (raw.astype(np.float32) * np.float32(1.0 / 2048)).view(np.complex64))
    p = tmp_path / "iq.bin"
    raw.tofile(p)
    np.testing.assert_array_equal(simpleBinRead(str(p)), raw.astype(np.float32).view(np.complex64))
    np.testing.assert_array_equal(simpleBinRead(str(p), numSamps=100, offset=40),
                                  raw[20:220].astype(np.float32).view(np.complex64))
    assert simpleBinReadToDevice(str(p), 10).shape == (10,)
    with pytest.raises(TypeError):
        simpleBinRead(str(p), in_dtype=np.complex64)
    with pytest.raises(TypeError):
        iq16_to_complex64(asarray(raw.astype(np.int32)))
    # multi-file and prefetching readers (usrpRoutines.py:70-156)
    import concurrent.futures

    from pydsproutines_amd.usrpRoutines import futureBinRead, multiBinRead, multiBinReadThreaded

    files = []
    want = []
    for i in range(3):
        r = rng.integers(-3000, 3000, 2 * 700, dtype=np.int16)
        q = tmp_path / ("f%d.bin" % i)
        r.tofile(q)
        files.append(str(q))
        want.append(r[:2 * 500].astype(np.float32).view(np.complex64))
    want = np.concatenate(want)
    np.testing.assert_array_equal(multiBinRead(files, 500), want)
    np.testing.assert_array_equal(multiBinReadThreaded(files, 500, threads=2), want)
    with concurrent.futures.ThreadPoolExecutor(max_workers=1) as ex:
        fut = futureBinRead(ex, files[1], 500)
        np.testing.assert_array_equal(iq16_to_complex64(asarray(fut.result())).get(), want[500:1000])
        with pytest.raises(TypeError):
            futureBinRead(ex, files[0], 10, in_dtype=np.complex64)


@pytest.mark.parametrize("dsr,phase,ntaps", [(1, 0, 64), (2, 1, 33), (4, 0, 128), (5, 3, 100), (8, 7, 257), (16, 5, 2048),
                                              (3, 2, 1)])
def test_iq16_frontend_fir_decimate(dsr, phase, ntaps):
    """ingest + FIR + decimation in one kernel == convert -> lfilter -> [phase::dsr] (filter.cu:9-58),
    one call and ragged streaming chunks; the same decimating kernel behind filter_smtaps(dsr=, dsPhase=)."""
    import scipy.signal as sps

    from pydsproutines_amd import asarray
    from pydsproutines_amd.filterRoutines import CupyKernelFilter
    from pydsproutines_amd.usrpRoutines import Iq16FrontEnd, iq16_to_complex64

    rng = np.random.default_rng(1000 * dsr + ntaps)
    n = 50_021
    raw = rng.integers(-2048, 2048, 2 * n, dtype=np.int16)
    taps = (sps.firwin(ntaps, 0.8 / dsr) if ntaps > 1 else np.array([0.75])).astype(np.float32)
    scale = 1.0 / 2048
    x = (raw.astype(np.float32) * np.float32(scale)).view(np.complex64)
    ref = sps.lfilter(taps.astype(np.float64), 1, x.astype(np.complex128))[phase::dsr]
    d_taps = asarray(taps)
    fe = Iq16FrontEnd(d_taps, dsr, phase, scale)
    got = fe.run(asarray(raw)).get()
    assert got.dtype == np.complex64 and got.size == ref.size
    np.testing.assert_allclose(got, ref, atol=2e-5)
    # the unfused chain through the reference-signature wrappers gives the same values
    chain = CupyKernelFilter().filter_smtaps(iq16_to_complex64(asarray(raw), scale), d_taps, dsr=dsr, dsPhase=phase).get()
    np.testing.assert_allclose(chain, ref, atol=2e-5)
    np.testing.assert_allclose(got, chain, atol=1e-6)
    # streaming: ragged chunks (shorter than the taps, not multiples of dsr) == the one-shot result
    fe.reset(phase)
    cuts = [0, 7, 8, 1000, 1003, 20_000, 20_001, 37_777, n]
    parts = [fe.run(asarray(raw[2 * a : 2 * b])).get() for a, b in zip(cuts[:-1], cuts[1:])]
    np.testing.assert_allclose(np.concatenate(parts), ref, atol=2e-5)


def test_iq16_frontend_validation():
    from pydsproutines_amd import asarray
    from pydsproutines_amd.usrpRoutines import Iq16FrontEnd

    taps = asarray(np.ones(8, np.float32))
    with pytest.raises(ValueError):
        Iq16FrontEnd(taps, 4, 4)
    with pytest.raises(TypeError):
        Iq16FrontEnd(asarray(np.ones(8, np.float64)))
    Iq16FrontEnd(taps, 17, 0)  # (round 2: any decimation factor / tap count -- the overlap-save form takes over)
    fe = Iq16FrontEnd(taps, 2)
    with pytest.raises(TypeError):
        fe.run(asarray(np.zeros(10, np.float32)))
    with pytest.raises(ValueError):
        fe.run(asarray(np.zeros(9, np.int16)))
