"""TemplateCrossCorrelator.correlate (complex64 (T, S) plane, or the per-delay (value, template) maxima) by template length: 16384-point
blocks up to 8192 samples, the chained roles beyond (complex-QF rows written by their own work items since round 5; the rocfft
engine before).  2^24-sample input; two untimed calls first (plans, rocFFT kernels, pool growth)."""
import sys
import time

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import cn, qpsk
from pydsproutines_amd import _lib, asarray
from pydsproutines_amd.xcorrRoutines import TemplateCrossCorrelator
rng = np.random.default_rng(1)
M = 1 << 24
d_x = asarray(cn(rng, M))
for T, L in ((64, 4096), (64, 12000), (16, 16384), (16, 30000)):
    tm = asarray(np.stack([qpsk(rng, L) for _ in range(T)]))
    tcc = TemplateCrossCorrelator(tm, M)
    for mode in (False, True):
        for _ in range(2):
            out = tcc.correlate(d_x, returnMax=mode)
        _lib.check(_lib.load().caf_stream_sync(None))
        t0 = time.perf_counter()
        for _ in range(3):
            out = tcc.correlate(d_x, returnMax=mode)
        _lib.check(_lib.load().caf_stream_sync(None))
        print("TCC T=%d L=%d returnMax=%s: %.2f ms per call" % (T, L, mode, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    del out, tcc
