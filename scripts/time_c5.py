"""Config C5 at full size: C2 coarse CAF (2^24 samples x 256 bins) + top-8 local maxima + CZT zoom per peak."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd import CAFPlan, _lib, asarray  # noqa: E402
from pydsproutines_amd.zoom import caf_with_zoom  # noqa: E402

N, M, F = 4096, 1 << 24, 256
fs = float(N)
rng = np.random.default_rng(5)
t = qpsk(rng, N)
rx = cn(rng, M)
truth = []
for i in range(8):
    d = 1_000_000 + 1_900_000 * i + 17 * i
    f = -100.0 + 27.3 * i
    rx[d : d + N] += ((1.0 - 0.05 * i) * t * np.exp(2j * np.pi * f * np.arange(N) / fs)).astype(np.complex64)
    truth.append((d, f))
d_rx = asarray(rx)
bins = np.arange(-F // 2, F // 2)
plan = CAFPlan(t, max_rx_len=M, bins=bins, grid=N)


def sync():
    _lib.check(_lib.load().caf_stream_sync(None))


for it in range(3):
    sync()
    t0 = time.perf_counter()
    res = plan.run(d_rx, surface=False, rows=True, peak=True)
    sync()
    t1 = time.perf_counter()
    out = caf_with_zoom(plan, d_rx, res, bins, N, fs, k=8, span_bins=1.0, step_bins=1.0 / 64)
    t2 = time.perf_counter()
    print("coarse CAF %.2f ms, top-k + zoom %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
found = sorted((o["delay"], round(o["fine_freq"], 3)) for o in out)
print("found", found)
assert [d for d, _ in found] == [d for d, _ in truth]
assert all(abs(f - g) <= 3.0 / 64 for (_, f), (_, g) in zip(found, truth))
