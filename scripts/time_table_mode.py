"""C2-sized job with the frequency hypotheses given as bins (one shifted spectrum) and as an explicit list of
normalised frequencies (one spectrum row per hypothesis): cost of the table mode."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd import CAFPlan, _lib, asarray  # noqa: E402

N, M, F = 4096, 1 << 24, 256
rng = np.random.default_rng(5)
t = qpsk(rng, N)
rx = cn(rng, M)
rx[1_000_000 : 1_000_000 + N] += t
d_rx = asarray(rx)
bins = np.arange(-F // 2, F // 2)


def sync():
    _lib.check(_lib.load().caf_stream_sync(None))


for name, kw in (("bins", dict(bins=bins, grid=N)), ("freqs_norm (same grid)", dict(freqs_norm=bins / N)),
                 ("freqs_norm (CZT-like grid, 201 bins)", dict(freqs_norm=(np.arange(201) - 100) * 0.1 / N))):
    for surf in (True, False):
        plan = CAFPlan(t, max_rx_len=M, **kw)
        res = plan.run(d_rx, surface=surf, rows=True, peak=True)
        sync()
        t0 = time.perf_counter()
        for _ in range(3):
            res = plan.run(d_rx, surface=surf, rows=True, peak=True, out=res)
        sync()
        dt = (time.perf_counter() - t0) / 3
        print("%-40s surface=%-5s %-10s %7.2f ms  peak %d" % (name, surf, plan.engine_used, dt * 1e3, int(res.peak_delay.get()[0])), flush=True)
        plan.close()
        del res
