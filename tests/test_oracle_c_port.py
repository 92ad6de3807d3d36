"""The plain-C, threaded restatement of the reference's native correlator (oracle/c/ippxcorrfft_port.c,
IppXcorrFFT.cpp:94-196) is pinned to the KAT-2 golden vector (generated from the imported reference's
fastXcorr branch B) and to the NumPy oracle, which is itself pinned to the reference."""

import numpy as np
import pytest

import oracle as O
from oracle import cport
from conftest import cn

TOL = 2e-5


def test_c_port_kat2(golden):
    g = golden("kat2_ippxcorrfft")
    pk, fi = cport.IppXcorrFFT(g["cutout"], num_threads=3).xcorr(g["data"], 0, 100, 3)
    assert pk.dtype == np.float32 and fi.dtype == np.int32 and pk.size == 34
    np.testing.assert_allclose(pk[:24], g["qf2"], atol=TOL)  # length 30: the direct-DFT branch
    np.testing.assert_array_equal(fi[:24], g["freqidx"])
    assert np.all(pk[24:] == 0) and np.all(fi[24:] == 0)  # IppXcorrFFT.cpp:125-130
    pk2, _ = cport.IppXcorrFFT(g["cutout"], 2).xcorr(g["data"], -6, 10, 3)
    assert np.all(pk2[:2] == 0) and pk2[2] == pytest.approx(g["qf2"][0], abs=TOL)


@pytest.mark.parametrize("n", [8, 64, 512, 2048, 4096])
def test_c_port_vs_numpy_oracle_pow2(n):
    """radix-4 / radix-2 Stockham branch (even and odd log2 n) against oracle.IppXcorrFFT, planted frequency offset."""
    rng = np.random.default_rng(100 + n)
    rx = cn(rng, 3 * n + 500)
    k0 = n // 8 + 1
    cut = (rx[200 : 200 + n] * np.exp(-2j * np.pi * k0 * np.arange(n) / n)).astype(np.complex64)
    ref_pk, ref_fi = O.IppXcorrFFT(cut, 1).xcorr(rx, 150, 260, 1)
    for threads in (1, 4, 7):
        pk, fi = cport.IppXcorrFFT(cut, threads).xcorr(rx, 150, 260, 1)
        np.testing.assert_allclose(pk, ref_pk, atol=TOL)
        clear = ref_pk > 0.5
        np.testing.assert_array_equal(fi[clear], ref_fi[clear])
    assert int(np.argmax(pk)) == 50 and fi[50] == k0 and abs(pk[50] - 1) < 1e-4


def test_c_port_argument_errors():
    rx = cn(np.random.default_rng(1), 100)
    lib = cport.load()
    pk, fi = np.zeros(5, np.float32), np.zeros(5, np.int32)
    cut = rx[:10].copy()
    # wrong output length -> the reference's std::runtime_error (IppXcorrFFT.cpp:63-66)
    assert lib.ippxcorrfft_port(cut.ctypes.data, 10, 1, rx.ctypes.data, 100, 0, 12, 2, 1, pk, fi, 5) == 1
    assert lib.ippxcorrfft_port(cut.ctypes.data, 10, 1, rx.ctypes.data, 100, 0, 10, 2, 1, pk, fi, 5) == 0
    with pytest.raises(ValueError):
        cport.IppXcorrFFT(rx.astype(np.complex128))
