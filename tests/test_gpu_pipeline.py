"""End to end on the GPU, the callers and data formats either side of the path: a raw interleaved
int16 IQ recording on disk (usrpRoutines.py:51-67 format) -> front-end FIR + decimate fused into the load ->
CAF over delay x frequency bins -> top-k local maxima -> chirp-Z zoom; every stage against the oracle chain
(convert -> scipy lfilter -> decimate -> the reference's per-delay CAF -> cztXcorr)."""

import numpy as np
import pytest
import scipy.signal as sps

import oracle as O
from conftest import cn, qpsk

pytestmark = pytest.mark.gpu


def test_iq_file_to_refined_peaks(tmp_path):
    from pydsproutines_amd import CAFPlan, asarray
    from pydsproutines_amd.usrpRoutines import Iq16FrontEnd
    from pydsproutines_amd.zoom import caf_with_zoom

    rng = np.random.default_rng(2024)
    dsr, n, fs = 2, 512, 512.0  # template defined at the decimated rate; bins 1 Hz wide
    m_raw = 240_000
    t = qpsk(rng, n)
    # wideband recording: noise + the template upsampled by 2 (zero-order hold) at two delays / offsets
    raw_c = 300.0 * cn(rng, m_raw)
    planted = [(30_000, 2.40, 900.0), (81_234, -5.70, 700.0)]  # (decimated delay, Hz, amplitude)
    for d, f, a in planted:
        seg = a * t * np.exp(2j * np.pi * f * np.arange(n) / fs)
        raw_c[dsr * d : dsr * (d + n)] += np.repeat(seg, dsr)
    raw = np.empty(2 * m_raw, np.int16)
    raw[0::2] = np.clip(np.round(raw_c.real), -32768, 32767)
    raw[1::2] = np.clip(np.round(raw_c.imag), -32768, 32767)
    path = tmp_path / "capture.bin"
    raw.tofile(path)

    # ---- device chain, the recording streamed in three ragged chunks
    taps = sps.firwin(64, 0.45).astype(np.float32)
    scale = 1.0 / 1024
    fe = Iq16FrontEnd(asarray(taps), dsr, 0, scale)
    disk = np.fromfile(path, dtype=np.int16)
    cuts = [0, 2 * 70_001, 2 * 160_000, disk.size]
    parts = [fe.run(asarray(disk[a:b])) for a, b in zip(cuts[:-1], cuts[1:])]
    rx = np.concatenate([p.get() for p in parts])

    # ---- oracle chain for the front end
    x = (disk.astype(np.float32) * np.float32(scale)).view(np.complex64)
    ref_rx = sps.lfilter(taps.astype(np.float64), 1, x.astype(np.complex128))[::dsr]
    assert rx.size == ref_rx.size == m_raw // dsr
    np.testing.assert_allclose(rx, ref_rx, atol=2e-5 * np.abs(ref_rx).max())

    # ---- CAF + zoom on the device
    bins = np.arange(-8, 8)
    d_rx = asarray(rx)
    plan = CAFPlan(t, max_rx_len=rx.size, bins=bins, grid=n)
    res = plan.run(d_rx, surface=True, rows=True, peak=True)
    out = caf_with_zoom(plan, d_rx, res, bins, n, fs, k=2, span_bins=1.0, step_bins=1.0 / 32)
    # the FIR delays the signal by (64 - 1) / 2 input samples ~ 16 decimated samples
    gd = int(round((taps.size - 1) / 2 / dsr))
    for o, (d, f, _) in zip(sorted(out, key=lambda o: o["delay"]), planted):
        assert abs(o["delay"] - (d + gd)) <= 1
        assert abs(o["fine_freq"] - f) <= 1.0 / 32 + 1e-9
        # the reference's per-delay CAF and CZT at that delay, on the oracle's filtered signal
        row = O.caf_bins(t, ref_rx.astype(np.complex64), bins, np.array([o["delay"]]))[0]
        np.testing.assert_allclose(res.surface.get()[0][o["delay"]], row, atol=1e-4 * row.max())
        refz, fr = O.cztXcorr(t, ref_rx.astype(np.complex64), o["coarse_bin"] - 1.0, o["coarse_bin"] + 1.0, fs,
                              cztStep=1.0 / 32, outputCAF=True, shifts=np.array([o["delay"]]))
        assert abs(refz[0].max() - o["fine_qf2"]) <= 2e-4
        assert abs(fr[int(np.argmax(refz[0]))] - o["fine_freq"]) <= 1e-9
    plan.close()
