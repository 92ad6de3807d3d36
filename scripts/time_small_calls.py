"""Latency of the functional host API at small sizes (config C1: one 1024-sample template against 65536 samples), where
plan creation, allocation and copies -- not kernels -- are what a call costs.  Median of repeated calls, NumPy in / out."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd.xcorrRoutines import fastXcorr  # noqa: E402

rng = np.random.default_rng(0)
n, m = 1024, 65536
t = qpsk(rng, n)
rx = cn(rng, m)
rx[3000 : 3000 + n] += t


def med(f, reps=9):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts)), 1e3 * float(np.min(ts))


for name, f in (("A  fastXcorr(cutout, rx)", lambda: fastXcorr(t, rx)),
                ("A' fastXcorr(absResult=False)", lambda: fastXcorr(t, rx, absResult=False)),
                ("B  fastXcorr(freqsearch=True), 2000 shifts", lambda: fastXcorr(t, rx, freqsearch=True, shifts=np.arange(2000, 4000))),
                ("C  fastXcorr(freqsearch, outputCAF), 500 shifts", lambda: fastXcorr(t, rx, freqsearch=True, outputCAF=True, shifts=np.arange(2800, 3300)))):
    a, b = med(f)
    print("%-52s median %7.2f ms  min %7.2f ms" % (name, a, b), flush=True)
