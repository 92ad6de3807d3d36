"""Timings of candidate plans (radix order / threads per row) of the run-time-compiled per-delay kernel, for the planner's cost model
(caf_jit.hip, pdj_cost).  One process per group of lengths may run beside others: timing sections are serialised by a file lock,
compilations are not.   usage: python scripts/sweep_pdj_plans.py OUT.csv N [N ...]"""
import ctypes as ct
import fcntl
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn  # noqa: E402
from pydsproutines_amd import _lib, asarray  # noqa: E402
from pydsproutines_amd.devarray import empty  # noqa: E402

RADICES = (25, 20, 18, 16, 15, 14, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2)
PT = 20


def multisets(n, max_r=25, depth=0):
    if n == 1:
        yield ()
        return
    if depth >= 5:
        return
    for r in RADICES:
        if r <= max_r and n % r == 0:
            for rest in multisets(n // r, r, depth + 1):
                yield (r,) + rest


def valid(n, rad, tpr):
    if len(rad) < 2 or tpr < 1 or tpr > 1024:
        return False
    for r in rad:
        cnt = -(-(n // r) // tpr)
        if cnt * r > max(PT, r):
            return False
    return True


def rough(n, rad, tpr):
    rpw = max(1, 256 // tpr)
    threads = ((rpw * tpr + 63) // 64 * 64) / rpw
    return threads * sum(-(-(n // r) // tpr) * r * 28 + 120 for r in rad)


def candidates(n, limit):
    out = set()
    for ms in multisets(n):
        if len(ms) < 2:
            continue
        cap = min(max(PT, r) // r * r for r in ms)
        t0 = -(-n // cap)
        for perm in set(itertools.permutations(ms)):
            for t in {t0, -(-t0 // 16) * 16, -(-t0 // 32) * 32, -(-t0 // 64) * 64}:
                if valid(n, perm, t):
                    out.add((perm, t))
    out = sorted(out, key=lambda c: rough(n, *c))
    rng = np.random.default_rng(n)
    head, tail = out[: limit * 2 // 3], out[limit * 2 // 3:]
    extra = [tail[i] for i in rng.choice(len(tail), min(len(tail), limit - len(head)), replace=False)] if tail else []
    return head + extra


if __name__ == "__main__":
    lib = _lib.load()
    out_path, lens = sys.argv[1], [int(a) for a in sys.argv[2:]]
    lock = open("/tmp/pdj_sweep.lock", "w")
    os.environ["CAF_JIT_ALL"] = "1"
    rng = np.random.default_rng(3)
    with open(out_path, "a") as f:
        for n in lens:
            num = 2_000_000 if n < 200 else (400_000 if n < 2000 else (200_000 if n < 6000 else 100_000))
            rx = cn(rng, n + num)
            d_rx, d_cut = asarray(rx), asarray(rx[500 : 500 + n].conj().copy())
            q, fi = empty(num, np.float32), empty(num, np.int32)
            p = lambda a: ct.c_void_p(a.ptr)  # noqa: E731

            def run():
                _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), rx.size, 0, 1, num, 0, p(q), p(fi), None, None, 0, None))

            for rad, tpr in candidates(n, int(os.environ.get("SWEEP_LIMIT", "48"))):
                plan = ",".join(map(str, rad)) + "/%d" % tpr
                os.environ["CAF_PDJ_PLAN"] = plan
                locked = False
                try:
                    # compiles (outside the lock) on a call of 64 rows: nothing that could disturb another process's timing
                    _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), rx.size, 0, 1, 64, 0, p(q), p(fi), None, None, 0, None))
                    _lib.check(lib.caf_stream_sync(None))
                    fcntl.flock(lock, fcntl.LOCK_EX)
                    locked = True
                    run()
                    _lib.check(lib.caf_stream_sync(None))
                    best = 1e9
                    for _ in range(3):
                        t0 = time.perf_counter()
                        for _ in range(3):
                            run()
                        _lib.check(lib.caf_stream_sync(None))
                        best = min(best, (time.perf_counter() - t0) / 3)
                    fcntl.flock(lock, fcntl.LOCK_UN)
                    locked = False
                    ok = int(np.argmax(q.get())) == 500
                except Exception as e:  # a plan the library rejects (LDS): recorded, not fatal
                    if locked:
                        fcntl.flock(lock, fcntl.LOCK_UN)
                    f.write("%d,%s,%d,nan,0,%s\n" % (n, "-".join(map(str, rad)), tpr, str(e)[:60].replace(",", ";")))
                    continue
                desc = ct.create_string_buffer(2048)
                lib.caf_perdelay_jit_describe(n, None, None, desc, 2048)
                d = dict(kv.split("=", 1) for kv in desc.value.decode().split(" ") if "=" in kv)
                f.write("%d,%s,%d,%.4f,%d,%s,%s,%s\n" % (n, "-".join(map(str, rad)), tpr, best * 1e3 * (100_000 / num), ok, d.get("rpw"),
                                                       d.get("conflict_cycles"), d.get("base_cycles")))
                f.flush()
            del d_rx
