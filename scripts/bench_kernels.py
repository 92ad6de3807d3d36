"""The reference's own kernel-level benchmark shapes (benchmarks/README.md,
benchmark_xcorrKernels.py, benchmark_multiTemplateDotKernels.py, benchmark_filterkernels.py,
benchmark_movAvgKernels.py) on one MI355X, through the reference-signature host layer, inputs resident on the
device where the reference's were.  The published numbers are for an unstated NVIDIA GPU."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn  # noqa: E402
from pydsproutines_amd import _lib, asarray  # noqa: E402
from pydsproutines_amd.cupyExtensions import multiTemplateSlidingDotProduct  # noqa: E402
from pydsproutines_amd.filterRoutines import CupyKernelFilter, cupyMovingAverage  # noqa: E402
from pydsproutines_amd.xcorrRoutines import cp_fastXcorr, cp_fastXcorr_v2  # noqa: E402

rng = np.random.default_rng(9)


def sync():
    _lib.check(_lib.load().caf_stream_sync(None))


def timeit(fn, reps=5):
    fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def report(name, t, published):
    print("%-78s %10.3f ms   (published: %s)" % (name, t * 1e3, published), flush=True)


# benchmarks/README.md:9-11,21-23 -- cp_fastXcorr, cutout N, 128 shifts, frequency search
for n, pub in ((1_000_000, "0.29 s cupy, 3.9 s C++/IPP 4 threads, 15.9 s NumPy"),
               (10_000_000, "3 s cupy, 62 s C++/IPP, 192 s NumPy")):
    rx = cn(rng, n + 1000)
    cut = rx[300 : 300 + n].copy()
    d_rx = asarray(rx)
    sh = np.arange(236, 364)
    t = timeit(lambda: cp_fastXcorr(cut, d_rx, shifts=sh), reps=2)
    report("cp_fastXcorr freqsearch, cutout %d, 128 shifts (incl. host copies of the results)" % n, t, pub)
    del d_rx

# benchmark_xcorrKernels.py:18-21 -- cp_fastXcorr_v2 chain, cutout 1000, 100 000 shifts
rx = cn(rng, 101_000)
d_rx, d_cut = asarray(rx), asarray(rx[5000:6000].copy())
t = timeit(lambda: cp_fastXcorr_v2(d_cut, d_rx, 0, 100_000, flattenCAF=True), reps=5)
report("cp_fastXcorr_v2 (product + FFT + argmax), cutout 1000, 100000 shifts", t, "3.05 ms for the product kernel alone")

# benchmark_multiTemplateDotKernels.py:18-19 -- 20 templates x 100, 10 M slides
x = cn(rng, 10_000_100)
d_x = asarray(x)
d_t = asarray(cn(rng, 20 * 100).reshape(20, 100))
t = timeit(lambda: multiTemplateSlidingDotProduct(d_x, d_t, 0, 10_000_000), reps=3)
report("multiTemplateSlidingDotProduct, 20 templates x 100, 10 M slides", t, "~260 ms")
del d_x

# benchmark_filterkernels.py:4-7 -- 128 taps on 1 M complex64
d_x = asarray(cn(rng, 1_000_000))
d_taps = asarray(rng.standard_normal(128).astype(np.float32))
f = CupyKernelFilter()
t = timeit(lambda: f.filter_smtaps(d_x, d_taps), reps=20)
report("FIR filter_smtaps, 128 taps, 1 M complex64", t, "317 us (smtaps), 230 us (sminput), 670 us cupy convolve")

# benchmark_movAvgKernels.py:5-7 -- moving average L = 100 on 10 M float32
d_f = asarray(rng.standard_normal(10_000_000).astype(np.float32))
t = timeit(lambda: cupyMovingAverage(d_f, 100), reps=20)
report("cupyMovingAverage, L = 100, 10 M float32", t, "472 us (kernel), 1.259 ms (FIR)")

# decimating FIR and the fused int16 front-end (no published figures: SURVEY 8f.2)
from pydsproutines_amd.usrpRoutines import Iq16FrontEnd, iq16_to_complex64  # noqa: E402

t = timeit(lambda: f.filter_smtaps(d_x, d_taps, dsr=4, dsPhase=1), reps=20)
report("FIR filter_smtaps dsr=4, 128 taps, 1 M complex64", t, "-")
raw = rng.integers(-2048, 2048, 2 * 16_777_216, dtype=np.int16)
d_raw = asarray(raw)
fe = Iq16FrontEnd(d_taps, 4, 0, 1.0 / 2048)


def fused():
    fe.reset(0)
    return fe.run(d_raw)


t = timeit(fused, reps=10)
report("int16 ingest + 128-tap FIR + decimate by 4 fused, 16 M samples", t, "-")
t = timeit(lambda: f.filter_smtaps(iq16_to_complex64(d_raw, 1.0 / 2048), d_taps, dsr=4), reps=10)
report("  same as convert kernel + decimating FIR kernel", t, "-")

# benchmark_czts.py:20-47 -- CZT of 10 rows x 10 000 samples onto 2001 bins (published: CPU, 3.0-4.7 ms per row)
from pydsproutines_amd.spectralRoutines import CZTCachedGPU  # noqa: E402

cz = CZTCachedGPU(10000, -1000.0, 1000.0, 1.0, 10000)
d_rows = asarray(cn(rng, 10 * 10000).reshape(10, 10000))
t = timeit(lambda: cz.runMany(d_rows), reps=20)
report("CZTCachedGPU.runMany, 10 rows x 10000 -> 2001 bins", t, "3.0-4.7 ms per row (CZTCached / scipy, CPU)")
