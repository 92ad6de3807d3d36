"""Config C3 through the reference's literal call: TemplateCrossCorrelator(64 templates x 4096, inputSize 2^24).correlate(x)
-> complex64 (64, 16 773 121) = 8.6 GB of output, and correlate(x, returnMax=True) (column maximum of that plane)."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import cn, qpsk
from pydsproutines_amd import _lib, asarray
from pydsproutines_amd.xcorrRoutines import TemplateCrossCorrelator
N, M, T = 4096, 1 << 24, 64
rng = np.random.default_rng(3)
tm = np.stack([qpsk(rng, N) for _ in range(T)])
rx = cn(rng, M)
for k in range(T):
    rx[100000 + 250000 * k : 100000 + 250000 * k + N] += tm[k]
d_rx = asarray(rx)
lib = _lib.load()
tcc = TemplateCrossCorrelator(asarray(tm), M)
out = tcc.correlate(d_rx)
_lib.check(lib.caf_stream_sync(None))
print("engine:", tcc._plan.engine_used, "block", tcc._plan.block)
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    out = tcc.correlate(d_rx)
    _lib.check(lib.caf_stream_sync(None))
    ts.append((time.perf_counter() - t0) * 1e3)
print("TCC.correlate(x) C3, complex plane: median %.2f ms, min %.2f ms (8.6 GB out = %.2f TB/s)" % (np.median(ts), min(ts), 8.588e9 / min(ts) / 1e9))
col = out.get()[:, 100000 + 250000 * 5]
assert np.argmax(np.abs(col)) == 5 and abs(abs(col[5]) - 0.7071) < 0.05, col[:8]
del out
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    qf, ti = tcc.correlate(d_rx, returnMax=True)
    _lib.check(lib.caf_stream_sync(None))
    ts.append((time.perf_counter() - t0) * 1e3)
print("TCC.correlate(x, returnMax=True) C3: median %.2f ms, min %.2f ms" % (np.median(ts), min(ts)))
assert int(ti.get()[100000 + 250000 * 7]) == 7
