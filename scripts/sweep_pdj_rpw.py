"""Rows per workgroup of the run-time-compiled per-delay kernel: the tuned plan of every swept length (csrc/caf_pdj_tuned.inc) timed
with 1 .. 8 rows per workgroup (workgroups of up to 640 threads) through CAF_PDJ_PLAN="radices/threads x rows".
usage: python scripts/sweep_pdj_rpw.py OUT.csv [N ...]"""
import ctypes as ct
import os
import re
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn  # noqa: E402
from pydsproutines_amd import _lib, asarray  # noqa: E402
from pydsproutines_amd.devarray import empty  # noqa: E402

lib = _lib.load()
tuned = {}
for line in open("pydsproutines_amd/csrc/caf_pdj_tuned.inc"):
    m = re.match(r'\{(\d+), "([\d,]+)/(\d+)(?:x(\d+))?"', line)
    if m:
        tuned[int(m.group(1))] = (m.group(2), int(m.group(3)))
lens = [int(a) for a in sys.argv[2:]] or sorted(tuned)
os.environ["CAF_JIT_ALL"] = "1"
rng = np.random.default_rng(3)
with open(sys.argv[1], "a") as f:
    for n in lens:
        rad, tpr = tuned[n]
        num = 2_000_000 if n < 200 else (400_000 if n < 2000 else (200_000 if n < 6000 else 100_000))
        rx = cn(rng, n + num)
        d_rx, d_cut = asarray(rx), asarray(rx[500 : 500 + n].conj().copy())
        q, fi = empty(num, np.float32), empty(num, np.int32)
        p = lambda a: ct.c_void_p(a.ptr)  # noqa: E731

        def run(rows=num):
            _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), rx.size, 0, 1, rows, 0, p(q), p(fi), None, None, 0, None))

        default = max(1, 256 // tpr)
        for rpw in sorted({r for r in range(1, 9) if r * tpr <= 640} | {default}):
            os.environ["CAF_PDJ_PLAN"] = "%s/%dx%d" % (rad, tpr, rpw)
            try:
                run(64)
                _lib.check(lib.caf_stream_sync(None))
                run()
                _lib.check(lib.caf_stream_sync(None))
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(3):
                        run()
                    _lib.check(lib.caf_stream_sync(None))
                    best = min(best, (time.perf_counter() - t0) / 3)
                ok = int(np.argmax(q.get())) == 500
            except Exception as e:
                f.write("%d,%s,%d,%d,nan,0,%s\n" % (n, rad.replace(",", "-"), tpr, rpw, str(e)[:60].replace(",", ";")))
                continue
            desc = ct.create_string_buffer(2048)
            lib.caf_perdelay_jit_describe(n, None, None, desc, 2048)
            d = dict(kv.split("=", 1) for kv in desc.value.decode().split(" ") if "=" in kv)
            f.write("%d,%s,%d,%d,%.4f,%d,%s,%s,%s%s\n" % (n, rad.replace(",", "-"), tpr, rpw, best * 1e3 * (100_000 / num), ok, d.get("rpw"), d.get("wg"),
                                                        d.get("lds_bytes"), ",default" if rpw == default else ""))
            f.flush()
        del d_rx
