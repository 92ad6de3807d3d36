import ctypes as ct, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import cn
from pydsproutines_amd import _lib, asarray
from pydsproutines_amd.devarray import empty
lib = _lib.load()
rng = np.random.default_rng(3)
for n, num in ((128, 2_000_000), (256, 1_000_000), (512, 1_000_000)):
    rx = cn(rng, n + num)
    d_rx, d_cut = asarray(rx), asarray(rx[500:500+n].conj().copy())
    q, fi = empty(num, np.float32), empty(num, np.int32)
    p = lambda a: ct.c_void_p(a.ptr)
    def run():
        _lib.check(lib.caf_xcorr_perdelay(p(d_cut), n, p(d_rx), rx.size, 0, 1, num, 0, p(q), p(fi), None, None, 0, None))
    run(); _lib.check(lib.caf_stream_sync(None))
    t0 = time.perf_counter()
    for _ in range(5): run()
    _lib.check(lib.caf_stream_sync(None))
    print("N=%4d x %d: %.3f ms" % (n, num, (time.perf_counter()-t0)/5*1e3), flush=True)
