"""Call latency of small jobs per engine (inputs resident, plan reused): is the one-launch engine also the
right default when there is little work?"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd import CAFPlan, _lib, asarray  # noqa: E402

rng = np.random.default_rng(1)


def sync():
    _lib.check(_lib.load().caf_stream_sync(None))


for n, m, f in ((1024, 65536, 1), (256, 8192, 32), (4096, 1 << 20, 256), (4096, 1 << 22, 256), (512, 1 << 20, 64)):
    t = qpsk(rng, n)
    d_rx = asarray(cn(rng, m))
    bins = np.arange(f) - f // 2
    line = "N=%5d M=%8d F=%3d:" % (n, m, f)
    for engine in ("persistent", "fused", "rocfft"):
        plan = CAFPlan(t, max_rx_len=m, bins=bins, grid=n, engine=engine)
        res = plan.run(d_rx, surface=True)
        sync()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            res = plan.run(d_rx, surface=True, out=res)
        sync()
        line += "  %s %.3f ms" % (engine, (time.perf_counter() - t0) / reps * 1e3)
        plan.close()
    print(line, flush=True)
