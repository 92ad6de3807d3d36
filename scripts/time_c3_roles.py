"""C3 (64 templates x 4096 samples, no frequency scan, 2^24-sample rx) on the persistent engine with the per-role clocks of
CAF_PERSIST_DEBUG=1: what a work item of the launch costs, by output (rows + peak, peak only) and item size."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd import CAFPlan, _lib, asarray  # noqa: E402

M, N, T = 1 << 24, 4096, 64
rng = np.random.default_rng(3)
tm = np.stack([qpsk(rng, N) for _ in range(T)])
rx = cn(rng, M)
rx[1_000_000 : 1_000_000 + N] += tm[7]
d_rx = asarray(rx)
sync = lambda: _lib.check(_lib.load().caf_stream_sync(None))  # noqa: E731
plan = CAFPlan(tm, max_rx_len=M, grid=N, bins=[0], engine="persistent")
for name, rk in (("rows + peak", dict(rows=True, peak=True)), ("peak only", dict(rows=False, peak=True)), ("rows only", dict(rows=True, peak=False))):
    res = plan.run(d_rx, **rk)
    sync()
    t0 = time.perf_counter()
    for _ in range(5):
        res = plan.run(d_rx, out=res, **rk)
    sync()
    dt = (time.perf_counter() - t0) / 5
    print("C3 %-12s %6.2f ms per pass (hyp_per_wg env %s)  peak7 = %d" % (name, dt * 1e3, os.environ.get("CAF_HYP_PER_WG", "-"),
                                                                       int(res.peak_delay.get()[7]) if res.peak_delay is not None else -1), flush=True)
plan.close()
