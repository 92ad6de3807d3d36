"""Calibration data for the plan cost model of the mixed-radix per-delay kernel (caf_perdelay_mr.hip, mr_cost): every
valid plan of a set of lengths (radices non-increasing, threads per row at the minimum and at the next multiples of 16 / 64),
timed through CAF_MR_PLAN.  Writes N, plan, tpr, ms lines; scripts/fit_mr_model.py fits the per-radix weights to them.
python scripts/sweep_mr_plans.py out.csv [N ...]"""
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

FIRST = [20, 18, 16, 15, 14, 12, 10, 9, 8, 7, 5]
LATER = [20, 16, 15, 10, 9, 8, 7, 6, 5, 4, 3, 2]
PT = 20


def plans(n):
    out = []

    def rec(rem, max_r, cur):
        if rem == 1:
            cap = min([16] + [PT // r * r for r in cur])
            t0 = -(-n // cap)
            cands = {t0, -(-t0 // 16) * 16, -(-t0 // 64) * 64}
            for t in sorted(cands):
                if t <= 1024 and t * 16 >= n and all(-(-(n // r) // t) <= PT // r for r in cur):
                    out.append((list(cur), t))
            return
        if len(cur) >= 6:
            return
        for r in (LATER if cur else FIRST):
            if r <= max_r and rem % r == 0:
                rec(rem // r, r, cur + [r])

    rec(n, 20, [])
    return out


def main():
    lib = _lib.load()
    p = lambda a: ct.c_void_p(a.ptr)  # noqa: E731
    out = open(sys.argv[1], "w")
    lengths = [int(a) for a in sys.argv[2:]] or [360, 600, 720, 900, 1200, 1400, 1500, 1920, 2400, 2800, 3000, 3600, 5000, 6000, 7000,
                                                 8000, 9600, 12000, 15000]
    rng = np.random.default_rng(3)
    num = 100000
    for n in lengths:
        rx = cn(rng, n + num)
        d_rx, d_cut = asarray(rx), asarray(rx[500 : 500 + n].conj().copy())
        q, fi = empty(num, np.float32), empty(num, np.int32)
        pls = plans(n)
        # (keep the sweep short: plans with at most one pass more than the shortest)
        shortest = min(len(r) for r, _ in pls)
        pls = [pl for pl in pls if len(pl[0]) <= shortest + 1]
        for rad, tpr in pls:
            os.environ["CAF_MR_PLAN"] = ",".join(map(str, rad)) + "/%d" % tpr
            best = 1e9
            for r in range(7):
                _lib.check(lib.caf_stream_sync(None))
                t0 = time.perf_counter()
                _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), rx.size, 0, 1, num, 0, p(q), p(fi), None, None, 0, None))
                _lib.check(lib.caf_stream_sync(None))
                if r >= 2:
                    best = min(best, time.perf_counter() - t0)
            ok = int(np.argmax(q.get())) == 500
            out.write("%d,%s,%d,%.4f,%d\n" % (n, "x".join(map(str, rad)), tpr, best * 1e3, ok))
            out.flush()
        print("N=%d: %d plans" % (n, len(pls)), flush=True)


main()
