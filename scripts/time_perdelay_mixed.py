"""Per-delay path, composite cutout lengths: the kernel compiled for the length at run time against the prebuilt plan-driven one
(CAF_JIT is read per call).  usage: python scripts/time_perdelay_mixed.py [N ...]"""
import ctypes as ct
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn  # noqa: E402
from pydsproutines_amd import _lib, asarray  # noqa: E402
from pydsproutines_amd.devarray import empty  # noqa: E402

lib = _lib.load()
rng = np.random.default_rng(3)
lens = [int(a) for a in sys.argv[1:]] or [1200, 1400, 5000, 1536, 1920, 3000, 3600, 8000, 12000, 96, 360]
for n in lens:
    num = 2_000_000 if n < 200 else (400_000 if n < 4000 else 100_000)  # (enough rows that the per-call work -- prefix, norm, launches -- is noise)
    rx = cn(rng, n + num)
    d_rx, d_cut = asarray(rx), asarray(rx[500 : 500 + n].conj().copy())
    q, fi = empty(num, np.float32), empty(num, np.int32)
    p = lambda a: ct.c_void_p(a.ptr)  # noqa: E731
    res = {}
    for jit in ("1", "0"):
        os.environ["CAF_JIT"] = jit

        def run():
            _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), rx.size, 0, 1, num, 0, p(q), p(fi), None, None, 0, None))

        run()
        _lib.check(lib.caf_stream_sync(None))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(3):
                run()
            _lib.check(lib.caf_stream_sync(None))
            best = min(best, (time.perf_counter() - t0) / 3)
        # a call of 1000 rows: what a call costs beside its rows
        t0 = time.perf_counter()
        for _ in range(10):
            _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), rx.size, 0, 1, 1000, 0, p(q), p(fi), None, None, 0, None))
        _lib.check(lib.caf_stream_sync(None))
        small = (time.perf_counter() - t0) / 10
        res[jit] = (best * 100_000 / num, int(np.argmax(q.get())), small)
    num = 100_000
    flops = num * 5.0 * n * np.log2(n)
    print("N=%5d per 1e5 delays: run-time kernel %7.3f ms (%5.1f TFLOP/s, %.3f of 157.3)   prebuilt %7.3f ms (%5.1f TFLOP/s)   a 1000-row call %.0f / %.0f us   peaks at %d / %d"
          % (n, res["1"][0] * 1e3, flops / res["1"][0] / 1e12, flops / res["1"][0] / 157.3e12, res["0"][0] * 1e3,
             flops / res["0"][0] / 1e12, res["1"][2] * 1e6, res["0"][2] * 1e6, res["1"][1], res["0"][1]), flush=True)
