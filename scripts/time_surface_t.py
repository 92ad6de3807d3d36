"""C2 (one 4096-sample template, 2^24-sample rx, 256 on-grid bins) by output: the delay-major surface (reference layout,
tile role), the hypothesis-major surface (caf_outputs.d_surface_t: rows written by the FFT work items), no surface."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd import CAFPlan, _lib, asarray  # noqa: E402

M, F, N = 1 << 24, 256, 4096
rng = np.random.default_rng(4)
rx = cn(rng, M)
t = qpsk(rng, N)
rx[5_000_000 : 5_000_000 + N] += t
d_rx = asarray(rx)
plan = CAFPlan(t, max_rx_len=M, bins=np.arange(-F // 2, F // 2), grid=N)
sync = lambda: _lib.check(_lib.load().caf_stream_sync(None))  # noqa: E731
for rep in range(2):
    for name, kw in (("delay-major surface", dict(surface=True)), ("hypothesis-major surface", dict(surface_t=True)),
                     ("hypothesis-major, S = 2^24 - 4096", dict(surface_t=True, num_shifts=M - N)),
                     ("hypothesis-major surface alone", dict(surface_t=True, rows=False, peak=False)), ("no surface", dict())):
        res = plan.run(d_rx, **kw)
        sync()
        t0 = time.perf_counter()
        for _ in range(10):
            res = plan.run(d_rx, out=res, **kw)
        sync()
        dt = (time.perf_counter() - t0) / 10
        pk = int(res.peak_delay.get()[0]) if res.peak_delay is not None else -1
        print("%-34s %7.2f ms per pass  %7.1f Mdelays/s  peak delay %d" % (name, dt * 1e3, (M - N + 1) / dt / 1e6, pk), flush=True)
        del res
plan.close()
