"""The C2 shape (2^24-sample rx, 256 on-grid bins, full surface) for template lengths around the LDS engines' limits:
N = 4096 / 8192 (16384-point blocks), 8193 / 16384 (32768-point blocks = two chained transforms), 16385 / 24576 / 32768
(65536-point blocks in the folded form: two chained transforms per output residue), 32769 (rocfft engine)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd import CAFPlan, _lib, asarray  # noqa: E402

M, F = 1 << 24, 256
rng = np.random.default_rng(4)
rx = cn(rng, M)
d_rx = asarray(rx)
LENGTHS = tuple(int(a) for a in sys.argv[1:]) or (4096, 8192, 8193, 16384, 16385, 24576, 32768, 32769)  # (optional: the lengths to time)
for n in LENGTHS:
    t = qpsk(rng, n)
    grid = min(16384, 1 << int(np.ceil(np.log2(n))))
    both = n > 8192 and (len(sys.argv) == 1 or os.environ.get("TIME_BOTH_ENGINES"))  # (in-LDS engine and the rocfft engine)
    for engine in (("auto", "rocfft") if both else ("auto",)):
        plan = CAFPlan(t, max_rx_len=M, bins=np.arange(-F // 2, F // 2), grid=grid, engine=engine)
        res = plan.run(d_rx, surface=True)
        _lib.check(_lib.load().caf_stream_sync(None))
        t0 = time.perf_counter()
        for _ in range(3):
            res = plan.run(d_rx, surface=True, out=res)
        _lib.check(_lib.load().caf_stream_sync(None))
        dt = (time.perf_counter() - t0) / 3
        print("N=%6d engine=%-10s block=%6d  %8.2f ms per pass  %7.1f Mdelays/s" % (n, plan.engine_used, plan.block, dt * 1e3, (M - n + 1) / dt / 1e6), flush=True)
        # the same plan without the surface: per-delay maxima + the peak only
        res2 = plan.run(d_rx, surface=False, rows=True, peak=True)
        _lib.check(_lib.load().caf_stream_sync(None))
        t0 = time.perf_counter()
        for _ in range(3):
            res2 = plan.run(d_rx, surface=False, rows=True, peak=True, out=res2)
        _lib.check(_lib.load().caf_stream_sync(None))
        dt = (time.perf_counter() - t0) / 3
        print("N=%6d engine=%-10s block=%6d  %8.2f ms per pass without the surface (per-delay maxima + peak)" % (n, plan.engine_used, plan.block, dt * 1e3), flush=True)
        del res2
        plan.close()
        del res
