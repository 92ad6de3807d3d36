"""Per-delay path (caf_xcorr_perdelay) before / after the fused kernel: cutout N x S delays, row results only.
Run once as is and once with CAF_PERDELAY_UNFUSED=1 (the switch is read once per process)."""
import ctypes as ct
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn  # noqa: E402
from pydsproutines_amd import _lib, asarray  # noqa: E402
from pydsproutines_amd.devarray import empty  # noqa: E402

lib = _lib.load()
rng = np.random.default_rng(3)
tag = "three-kernel form" if os.environ.get("CAF_PERDELAY_UNFUSED") == "1" else "fused kernel"
for n, num in ((4096, 1_000_000), (2048, 1_000_000), (8192, 200_000), (1024, 1_000_000), (256, 1_000_000), (16384, 100_000), (1000, 1_000_000), (1000, 100_000), (100, 1_000_000), (10000, 100_000), (1200, 100_000), (1536, 100_000), (3000, 100_000), (5000, 100_000), (12000, 100_000), (96, 1_000_000)):
    rx = cn(rng, n + num)
    d_rx, d_cut = asarray(rx), asarray(rx[500 : 500 + n].conj().copy())
    q, fi = empty(num, np.float32), empty(num, np.int32)
    p = lambda a: ct.c_void_p(a.ptr)  # noqa: E731

    def run():
        _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), rx.size, 0, 1, num, 0, p(q), p(fi), None, None, 0, None))

    run()
    _lib.check(lib.caf_stream_sync(None))
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        run()
    _lib.check(lib.caf_stream_sync(None))
    dt = (time.perf_counter() - t0) / reps
    flops = num * 5.0 * n * np.log2(n)
    print("%-18s N=%5d  %8d delays  %9.2f ms  %7.1f Mdelays/s  %6.1f GB/s of product elements (8 B each)  %5.1f TFLOP/s FFT  peak at %d"
          % (tag, n, num, dt * 1e3, num / dt / 1e6, num * n * 8 / dt / 1e9, flops / dt / 1e12, int(np.argmax(q.get()))), flush=True)
    del d_rx
