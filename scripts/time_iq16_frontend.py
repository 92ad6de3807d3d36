"""The int16 front end (usrpRoutines.Iq16FrontEnd: raw IQ -> FIR -> decimate in one kernel, k_fir_poly / k_fir_decim) on 2^24
samples by tap count and decimation factor."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pydsproutines_amd import _lib, asarray  # noqa: E402
from pydsproutines_amd.usrpRoutines import Iq16FrontEnd  # noqa: E402

n = 1 << 24
rng = np.random.default_rng(0)
d_iq = asarray(rng.integers(-2000, 2000, 2 * n).astype(np.int16))
sync = lambda: _lib.check(_lib.load().caf_stream_sync(None))  # noqa: E731
for ntaps, dsr in ((64, 4), (64, 2), (128, 4), (128, 8), (32, 4), (64, 16)):
    fe = Iq16FrontEnd(asarray((rng.standard_normal(ntaps) / 8).astype(np.float32)), dsr=dsr, scale=1.0 / 2048)
    fe.run(d_iq)
    sync()
    t0 = time.perf_counter()
    for _ in range(20):
        fe.run(d_iq)
    sync()
    dt = (time.perf_counter() - t0) / 20
    print("int16 front end  taps=%4d dsr=%2d  %7.1f us  %6.1f GB/s (4 B in + 8 B out per kept sample)" % (
        ntaps, dsr, dt * 1e6, (n * 4.0 + n / dsr * 8.0) / dt / 1e9), flush=True)
