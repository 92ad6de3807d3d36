"""One per-delay shape, repeated: python scripts/time_perdelay_one.py N num [reps]  (for rocprofv3 --kernel-trace --stats)."""
import ctypes as ct
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn  # noqa: E402
from pydsproutines_amd import _lib, asarray  # noqa: E402
from pydsproutines_amd.devarray import empty  # noqa: E402

n, num = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
lib = _lib.load()
rng = np.random.default_rng(3)
rx = cn(rng, n + num)
d_rx, d_cut = asarray(rx), asarray(rx[500 : 500 + n].conj().copy())
q, fi = empty(num, np.float32), empty(num, np.int32)
p = lambda a: ct.c_void_p(a.ptr)  # noqa: E731
for r in range(reps + 1):
    if r == 1:
        _lib.check(lib.caf_stream_sync(None))
        t0 = time.perf_counter()
    _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), rx.size, 0, 1, num, 0, p(q), p(fi), None, None, 0, None))
_lib.check(lib.caf_stream_sync(None))
print("N=%d x %d delays: %.3f ms per call, peak at %d" % (n, num, (time.perf_counter() - t0) / reps * 1e3, int(np.argmax(q.get()))))
