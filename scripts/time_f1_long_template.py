"""One hypothesis (no frequency search: fastXcorr branch A, TemplateCrossCorrelator) with templates beyond 8192 samples: the chained
roles of the in-LDS engine against the rocfft engine, rows + peak, 2^22-sample rx.   usage: python scripts/time_f1_long_template.py [N ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd import CAFPlan, _lib, asarray  # noqa: E402

M = 1 << 22
rng = np.random.default_rng(3)
d_rx = asarray(cn(rng, M))
for n in tuple(int(a) for a in sys.argv[1:]) or (12000, 20000, 32768, 50000, 100000):
    t = qpsk(rng, n)
    grid = 1 << int(np.ceil(np.log2(n)))
    for engine in ("auto", "rocfft"):
        plan = CAFPlan(t, max_rx_len=M, bins=[0], grid=grid, engine=engine)
        res = plan.run(d_rx, surface=False, rows=True, peak=True)
        _lib.check(_lib.load().caf_stream_sync(None))
        t0 = time.perf_counter()
        for _ in range(10):
            res = plan.run(d_rx, surface=False, rows=True, peak=True, out=res)
        _lib.check(_lib.load().caf_stream_sync(None))
        dt = (time.perf_counter() - t0) / 10
        print("N=%6d F=1 engine=%-10s block=%7d  %7.3f ms per call" % (n, plan.engine_used, plan.block, dt * 1e3), flush=True)
        plan.close()
