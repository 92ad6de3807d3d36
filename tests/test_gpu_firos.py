"""GPU parity tests of the overlap-save FIR (csrc/caf_firos.hip) behind caf_fir_lfilter / caf_iq16_fir_decimate
against scipy.signal.lfilter through oracle.kernels.filter_lfilter (the semantics the reference pins its own
kernels to, filterRoutines.py:1256): 128 taps (direct form, for reference), 300 / 1024 / 8192 taps (fused in-LDS
blocks of 1024 / 4096 / 16384) and 65 536 taps (rocFFT rows), carried-in history, decimation phases, chunked
streaming == one long filter, and the raw-int16 front end."""

import numpy as np
import pytest

from conftest import cn
from oracle import kernels as OK

pytestmark = pytest.mark.gpu


def _taps(rng, n):
    t = rng.standard_normal(n).astype(np.float32)
    return (t / np.sqrt(n)).astype(np.float32)  # unit-energy filter: outputs stay O(1)


@pytest.mark.parametrize("ntaps,n", [(128, 50_000), (300, 50_000), (1024, 70_000), (8192, 100_000), (65_536, 150_000)])
def test_lfilter_long_taps(ntaps, n):
    from pydsproutines_amd import asarray
    from pydsproutines_amd.filterRoutines import CupyKernelFilter

    rng = np.random.default_rng(ntaps)
    x = cn(rng, n)
    taps = _taps(rng, ntaps)
    f = CupyKernelFilter()
    got = f.filter_smtaps(asarray(x), asarray(taps)).get()
    ref = OK.filter_lfilter(x, taps)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 2e-5 * max(1.0, np.abs(ref).max())
    # decimated outputs are exactly the kept samples of the full-rate result (same kernel family: tolerance only)
    for dsr, ph in ((3, 1), (16, 15)):
        gd = f.filter_smtaps(asarray(x), asarray(taps), dsr=dsr, dsPhase=ph).get()
        rd = ref[ph::dsr]  # == OK.filter_lfilter(x, taps, dsr=dsr, dsPhase=ph): one scipy.signal.lfilter pass instead of three
        assert gd.shape == rd.shape
        assert np.max(np.abs(gd - rd)) <= 2e-5 * max(1.0, np.abs(rd).max())


@pytest.mark.parametrize("ntaps", [300, 8192, 20_000])
def test_streaming_state_equals_one_long_filter(ntaps):
    """run_filter_smtaps carries the last `memory` samples into the next call (filterRoutines.py:482-501)."""
    from pydsproutines_amd import asarray
    from pydsproutines_amd.filterRoutines import CupyKernelFilter

    rng = np.random.default_rng(ntaps + 1)
    x = cn(rng, 3 * 60_000)
    taps = _taps(rng, ntaps)
    ref = OK.filter_lfilter(x, taps)
    f = CupyKernelFilter(memory=ntaps - 1)
    d_t = asarray(taps)
    parts = [f.run_filter_smtaps(asarray(x[i * 60_000 : (i + 1) * 60_000]), d_t).get() for i in range(3)]
    got = np.concatenate(parts)
    assert np.max(np.abs(got - ref)) <= 2e-5 * max(1.0, np.abs(ref).max())
    # explicit carried-in history shorter than ntaps - 1: the missing part is zeros
    hist = cn(rng, 100)
    f2 = CupyKernelFilter(memory=100)
    from pydsproutines_amd import _lib
    import ctypes as ct
    _lib.check(_lib.load().caf_h2d(ct.c_void_p(f2.delay.ptr), hist.ctypes.data, hist.nbytes, None))
    g2 = f2.filter_smtaps(asarray(x[:50_000]), d_t, useInternalDelay=True).get()
    r2 = OK.filter_lfilter(x[:50_000], taps, delay=hist)
    assert np.max(np.abs(g2 - r2)) <= 2e-5 * max(1.0, np.abs(r2).max())


@pytest.mark.parametrize("ntaps,dsr", [(1500, 4), (3000, 1), (9000, 5), (200, 32)])
def test_iq16_front_end_long_taps(ntaps, dsr):
    """Raw int16 IQ -> filter -> decimate with more taps (or a larger factor) than the direct polyphase kernel takes,
    in chunks whose lengths are not multiples of dsr: equals one lfilter over the converted samples."""
    from pydsproutines_amd import asarray
    from pydsproutines_amd.usrpRoutines import Iq16FrontEnd

    rng = np.random.default_rng(ntaps + dsr)
    n = 150_000
    iq = rng.integers(-2000, 2000, 2 * n).astype(np.int16)
    scale = 1.0 / 2048
    x = (iq.astype(np.float32) * np.float32(scale)).view(np.complex64)
    taps = _taps(rng, ntaps)
    fe = Iq16FrontEnd(asarray(taps), dsr=dsr, dsPhase=dsr - 1, scale=scale)
    cuts = [0, 40_001, 40_001 + 59_998, n]
    got = np.concatenate([fe.run(asarray(iq[2 * a : 2 * b])).get() for a, b in zip(cuts[:-1], cuts[1:])])
    ref = OK.filter_lfilter(x, taps, dsr=dsr, dsPhase=dsr - 1)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 2e-5 * max(1.0, np.abs(ref).max())
