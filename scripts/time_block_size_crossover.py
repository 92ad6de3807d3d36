"""C2 shape for template lengths between 4096 and 8192: 16384-point blocks (default) against 32768-point blocks
(CAF_FUSED_LB15=1, read once per process: run the script once with and once without)."""
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
d_rx = asarray(cn(rng, M))
for n in (4096, 5120, 5632, 6144, 6656, 7168, 8192):
    t = qpsk(rng, n)
    for surface in (True, False):
        plan = CAFPlan(t, max_rx_len=M, bins=np.arange(-F // 2, F // 2), grid=8192 if n > 4096 else 4096)
        kw = dict(surface=True) if surface else dict(surface=False, rows=True, peak=True)
        res = plan.run(d_rx, **kw)
        _lib.check(_lib.load().caf_stream_sync(None))
        t0 = time.perf_counter()
        for _ in range(3):
            res = plan.run(d_rx, out=res, **kw)
        _lib.check(_lib.load().caf_stream_sync(None))
        dt = (time.perf_counter() - t0) / 3
        print("LB15=%s N=%5d block=%6d %-10s %8.2f ms per pass" % (os.environ.get("CAF_FUSED_LB15", "0"), n, plan.block,
                                                                  "surface" if surface else "no surface", dt * 1e3), flush=True)
        plan.close()
        del res
