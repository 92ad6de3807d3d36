"""Wall clock of whole reference-signature calls (host arrays in, host arrays out; plan creation included)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd.xcorrRoutines import GroupXcorr, cztXcorr, fastXcorr  # noqa: E402

rng = np.random.default_rng(0)


def timeit(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


rx = cn(rng, 65536)
cut = rx[1000:2024].copy()
print("C1 fastXcorr(cutout 1024, rx 65536), all 64513 delays: %.2f ms (reference NumPy loop: ~630 ms)" % (
    1e3 * timeit(lambda: fastXcorr(cut, rx))), flush=True)
rx2 = cn(rng, 1 << 20)
t = qpsk(rng, 4096)
print("fastXcorr(cutout 4096, rx 2^20), all delays: %.2f ms (reference: 9.8 us/shift -> ~10 s)" % (
    1e3 * timeit(lambda: fastXcorr(t, rx2))), flush=True)
sh = np.arange(0, 2000)
print("fastXcorr(freqsearch=True), 2000 shifts x 4096 bins: %.2f ms (reference: 56 us/shift -> 112 ms)" % (
    1e3 * timeit(lambda: fastXcorr(t, rx2, freqsearch=True, shifts=sh))), flush=True)
g = GroupXcorr(t[:2048], np.array([0, 1024]), np.array([1024, 1024]), np.arange(-128, 128) * 1.0, 4096.0)
print("GroupXcorr 2 groups x 1024, 256 freqs, 2000 shifts: %.2f ms (reference: 137 us/shift -> 274 ms)" % (
    1e3 * timeit(lambda: g.xcorr(rx2, sh))), flush=True)
print("cztXcorr 4096 cutout, 129 CZT bins, 2000 shifts: %.2f ms" % (
    1e3 * timeit(lambda: cztXcorr(t, rx2, -1.0, 1.0, 4096.0, 1.0 / 64, True, sh))), flush=True)
