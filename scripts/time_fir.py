"""FIR (caf_fir_lfilter through CupyKernelFilter.filter_smtaps): direct form vs overlap-save, 2^24 complex64 samples.
CAF_FIR_OS_MIN_TAPS=1000000 forces the direct form where it exists (<= 4096 taps); the switch is read once."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn  # noqa: E402
from pydsproutines_amd import _lib, asarray  # noqa: E402
from pydsproutines_amd.filterRoutines import CupyKernelFilter  # noqa: E402

rng = np.random.default_rng(2)
n = 1 << 24
d_x = asarray(cn(rng, n))
f = CupyKernelFilter()
forced = os.environ.get("CAF_FIR_OS_MIN_TAPS")
for ntaps in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536):
    if forced and int(forced) > 4096 and ntaps > 4096:
        continue
    d_t = asarray((rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(np.float32))
    for dsr in (1, 4):
        out = f.filter_smtaps(d_x, d_t, dsr=dsr)
        _lib.check(_lib.load().caf_stream_sync(None))
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            out = f.filter_smtaps(d_x, d_t, dsr=dsr)
        _lib.check(_lib.load().caf_stream_sync(None))
        dt = (time.perf_counter() - t0) / reps
        alg = n * 8 + (n // dsr) * 8  # 8 B read per input + 8 B written per output
        print("%-22s taps=%6d dsr=%d  %8.3f ms  %7.1f GB/s algorithmic (of 8000)  %6.1f Msamples/s in" % (
            ("direct forced" if int(forced) > 4096 else "overlap-save forced") if forced else "default dispatch", ntaps, dsr, dt * 1e3, alg / dt / 1e9, n / dt / 1e6), flush=True)

# upfirdn_sm (benchmark_upfirdnkernels.py's shape scaled up: 128 firwin taps, up 5/7/11, down 2/3)
if not forced:
    rows, ln = 64, 1 << 18
    d_m = asarray(cn(rng, rows * ln).reshape(rows, ln))
    d_t = asarray((rng.standard_normal(128) / np.sqrt(128)).astype(np.float32))
    for up, down in ((5, 2), (11, 3), (1, 1), (2, 7)):
        out = f.upfirdn_sm(d_m, d_t, up, down)
        _lib.check(_lib.load().caf_stream_sync(None))
        t0 = time.perf_counter()
        for _ in range(3):
            out = f.upfirdn_sm(d_m, d_t, up, down)
        _lib.check(_lib.load().caf_stream_sync(None))
        dt = (time.perf_counter() - t0) / 3
        alg = rows * ln * 8 + out.size * 8
        print("upfirdn_sm %d x %d, 128 taps, up %d down %d: %8.3f ms  %7.1f GB/s algorithmic" % (rows, ln, up, down, dt * 1e3, alg / dt / 1e9),
              flush=True)
