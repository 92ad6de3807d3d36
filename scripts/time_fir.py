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
            "direct forced" if forced else "default dispatch", ntaps, dsr, dt * 1e3, alg / dt / 1e9, n / dt / 1e6), flush=True)
