"""Several templates, no frequency search (config C3's kind of call) by template length: per-delay rows + peaks and peaks only, 2^24-sample rx:
16384-point blocks write the rows from the FFT items, the chained roles go through |y|^2 tiles."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import cn, qpsk
from pydsproutines_amd import CAFPlan, _lib, asarray
rng = np.random.default_rng(1)
M = 1 << 24
d_x = asarray(cn(rng, M))
lib = _lib.load()
for T, L in ((64, 4096), (64, 12000), (64, 16384), (16, 30000)):
    tm = np.stack([qpsk(rng, L) for _ in range(T)])
    plan = CAFPlan(tm, max_rx_len=M, bins=[0], grid=1 << int(np.ceil(np.log2(L))))
    for kw in (dict(surface=False, rows=True, peak=True), dict(surface=False, rows=False, peak=True)):
        res = None
        for _ in range(2):
            res = plan.run(d_x, out=res, **kw)
        _lib.check(lib.caf_stream_sync(None))
        t0 = time.perf_counter()
        for _ in range(5):
            res = plan.run(d_x, out=res, **kw)
        _lib.check(lib.caf_stream_sync(None))
        print("T=%d L=%d F=1 %s block=%d: %.2f ms" % (T, L, "rows+peak" if kw["rows"] else "peak only", plan.block, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    plan.close()
