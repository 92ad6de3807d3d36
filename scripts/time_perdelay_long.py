"""Per-delay path, cutouts longer than one LDS image (20001 .. 160000 samples): the run-time-compiled split kernel (CAF_JIT=1)
against product rows -> rocFFT rows -> argmax through HBM (CAF_JIT=0).  usage: python scripts/time_perdelay_long.py [N ...]"""
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
lens = [int(a) for a in sys.argv[1:]] or [24000, 32768, 40000, 50000, 65536, 80000, 100000, 131072]
for n in lens:
    num = 20_000
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
            run()
            _lib.check(lib.caf_stream_sync(None))
            best = min(best, time.perf_counter() - t0)
        qq = q.get()
        res[jit] = (best * 1e3, int(np.argmax(qq)), float(qq[500]), int(fi.get()[500]))
    flops = num * 5.0 * n * np.log2(n)
    desc = ct.create_string_buffer(2048)
    lib.caf_perdelay_jit_describe(n, None, None, desc, 2048)
    print("N=%6d x %d delays: run-time kernel %8.3f ms (%5.1f TFLOP/s, %.3f of 157.3)   rows through rocFFT %8.3f ms (%5.1f TFLOP/s)   "
          "peak at %d / %d value %.6f / %.6f bin %d / %d | %s"
          % (n, num, res["1"][0], flops / res["1"][0] / 1e9, flops / res["1"][0] / 1e9 / 157.3, res["0"][0], flops / res["0"][0] / 1e9,
             res["1"][1], res["0"][1], res["1"][2], res["0"][2], res["1"][3], res["0"][3], desc.value.decode()[:90]), flush=True)
    del d_rx
