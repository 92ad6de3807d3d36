"""Config C3 through the complex-QF path of the rocfft engine (TemplateCrossCorrelator's exact mode) for several
block sizes: 64 templates x 4096 vs 2^24 samples, complex64 (T, S) output."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd import CAFPlan, _lib, asarray  # noqa: E402

N, M, T = 4096, 1 << 24, 64
rng = np.random.default_rng(5)
tm = np.stack([qpsk(rng, N) for _ in range(T)])
d_rx = asarray(cn(rng, M))


def sync():
    _lib.check(_lib.load().caf_stream_sync(None))


for lb in (0, 14, 15, 16, 17, 18, 20):
    for nb in (0, 1, 4):
        try:
            plan = CAFPlan(tm, max_rx_len=M, bins=[0], grid=N, engine="rocfft", log2_block=lb, blocks_per_batch=nb)
        except (ValueError, MemoryError, RuntimeError) as e:
            print("log2_block=%d nb=%d n/a: %s" % (lb, nb, e))
            continue
        res = plan.run(d_rx, rows=False, peak=False, cqf=True)
        sync()
        t0 = time.perf_counter()
        for _ in range(2):
            res = plan.run(d_rx, rows=False, peak=False, cqf=True, out=res)
        sync()
        print("log2_block=%2d (used %d) blocks_per_batch=%d (used %d): %.1f ms" % (
            lb, plan.block, nb, plan.blocks_per_batch, (time.perf_counter() - t0) / 2 * 1e3), flush=True)
        plan.close()
        del res
