"""Mixed-radix per-delay kernel: the planner's choice against forced plans (CAF_MR_PLAN), per length.
python scripts/time_mr_plans.py  ->  ms per call of caf_xcorr_perdelay (row results only), 10 calls after a warm-up."""
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
p = lambda a: ct.c_void_p(a.ptr)  # noqa: E731


def run(n, num, plan, reps=10):
    if plan:
        os.environ["CAF_MR_PLAN"] = plan
    else:
        os.environ.pop("CAF_MR_PLAN", None)
    rng = np.random.default_rng(3)
    rx = cn(rng, n + num)
    d_rx, d_cut = asarray(rx), asarray(rx[500 : 500 + n].conj().copy())
    q, fi = empty(num, np.float32), empty(num, np.int32)
    for r in range(reps + 1):
        if r == 1:
            _lib.check(lib.caf_stream_sync(None))
            t0 = time.perf_counter()
        if r == 0:
            os.environ["CAF_MR_DEBUG"] = "1"
        else:
            os.environ.pop("CAF_MR_DEBUG", None)
        _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), rx.size, 0, 1, num, 0, p(q), p(fi), None, None, 0, None))
    _lib.check(lib.caf_stream_sync(None))
    ms = (time.perf_counter() - t0) / reps * 1e3
    ok = int(np.argmax(q.get())) == 500
    print("N=%6d x %8d  %-22s %8.3f ms %s" % (n, num, plan or "(planner)", ms, "" if ok else " WRONG PEAK"), flush=True)


CASES = [
    (1200, 100000, [None, "16,5,5,3/75", "16,5,5,3/80", "15,10,8/80", "20,10,6/67", "20,10,6/75", "12,10,10/100", "15,16,5/80"]),
    (1536, 100000, [None, "16,16,3,2/96", "16,16,6/96", "16,8,6,2/96", "12,16,8/128"]),
    (3000, 100000, [None, "10,10,10,3/188", "15,10,10,2/200", "20,15,10/200", "20,10,5,3/188", "15,20,10/200"]),
    (5000, 100000, [None, "10,10,10,5/313", "20,10,5,5/313", "20,10,5,5/320", "20,20,5,5,1/313"]),
    (12000, 100000, [None, "16,10,5,5,3/750", "20,15,10,4/800", "20,20,10,3/750", "16,15,10,5/800", "20,20,6,5/750"]),
    (96, 1000000, [None, "16,3,2/6", "16,6/6", "12,8/8"]),
    (1400, 100000, [None, "10,10,7,2/100", "14,10,10/100", "20,10,7/100"]),
    (7000, 100000, [None, "20,10,7,5/500", "14,10,10,5/500"]),
    (1920, 100000, [None, "16,15,8/128", "16,8,5,3/120", "20,16,6/120"]),
]
only = [int(a) for a in sys.argv[1:]]
for n, num, plans in CASES:
    if only and n not in only:
        continue
    for pl in plans:
        run(n, num, pl)
