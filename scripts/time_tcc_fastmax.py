import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import cn, qpsk
from pydsproutines_amd import _lib, asarray
from pydsproutines_amd.xcorrRoutines import TemplateCrossCorrelator
N, M, T = 4096, 1 << 24, 64
rng = np.random.default_rng(3)
tm = np.stack([qpsk(rng, N) for _ in range(T)])
d_rx = asarray(cn(rng, M))
tcc = TemplateCrossCorrelator(asarray(tm), M, fastMax=True)
lib = _lib.load()
qf, ti = tcc.correlate(d_rx, returnMax=True)
_lib.check(lib.caf_stream_sync(None))
t0 = time.perf_counter()
for _ in range(5):
    qf, ti = tcc.correlate(d_rx, returnMax=True)
_lib.check(lib.caf_stream_sync(None))
print("TCC fastMax C3: %.2f ms per call" % ((time.perf_counter() - t0) / 5 * 1e3))
