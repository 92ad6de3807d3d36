"""fastXcorr (host arrays in and out, plan creation included) over cutout lengths that cross the engines' boundaries: branch A (all
delays, no frequency search), A' (complex), B (per-delay maximum over the FFT grid, 2000 delays).  2^20-sample rx."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn  # noqa: E402
from pydsproutines_amd.xcorrRoutines import fastXcorr  # noqa: E402

rng = np.random.default_rng(0)
rx = cn(rng, 1 << 20)


def timeit(fn, reps=3):
    fn()
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return 1e3 * (time.perf_counter() - t0) / reps


for n in (1024, 4096, 8192, 8193, 12000, 16384, 16385, 30000, 32768, 32769, 65536, 100000, 262144, 262145, 400000):
    cut = rx[777 : 777 + n].copy()
    sh = np.arange(500, 2500)
    a = timeit(lambda: fastXcorr(cut, rx))
    ac = timeit(lambda: fastXcorr(cut, rx, absResult=False))
    b = timeit(lambda: fastXcorr(cut, rx, freqsearch=True, shifts=sh))
    print("N=%7d  A %7.2f ms   A' %7.2f ms   B(2000 delays) %8.2f ms" % (n, a, ac, b), flush=True)
