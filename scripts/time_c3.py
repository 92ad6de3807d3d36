import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import cn, qpsk
from pydsproutines_amd import CAFPlan, _lib, asarray
N, M, T = 4096, 1 << 24, 64
rng = np.random.default_rng(3)
tm = np.stack([qpsk(rng, N) for _ in range(T)])
d_rx = asarray(cn(rng, M))
plan = CAFPlan(tm, max_rx_len=M, bins=[0], grid=N)
lib = _lib.load()
for rows, surf in ((True, False), (True, True)):
    res = plan.run(d_rx, surface=surf, rows=rows, peak=True)
    _lib.check(lib.caf_stream_sync(None))
    t0 = time.perf_counter()
    for _ in range(5):
        res = plan.run(d_rx, surface=surf, rows=rows, peak=True, out=res)
    _lib.check(lib.caf_stream_sync(None))
    print("C3 rows=%d surface=%d: %.2f ms per pass" % (rows, surf, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
