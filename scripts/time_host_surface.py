"""Host-returning surface calls at the C2 shape (one 4096-sample template, 2^24-sample rx, 256 bins, every delay): the
(delays, frequencies) float64 array of cztXcorr(outputCAF=True) / GroupXcorrCZT.xcorr -- 34 GB on the host.
usage: python scripts/time_host_surface.py [new|old]   (old = CAF_HOST_SURFACE_DELAY_MAJOR=1: rounds 1-4's path)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
mode = sys.argv[1] if len(sys.argv) > 1 else "new"
if mode == "old":
    os.environ["CAF_HOST_SURFACE_DELAY_MAJOR"] = "1"
from conftest import cn, qpsk  # noqa: E402
import pydsproutines_amd.xcorrRoutines as X  # noqa: E402
from pydsproutines_amd import CAFPlan, _lib  # noqa: E402
from pydsproutines_amd.devarray import asarray  # noqa: E402

rng = np.random.default_rng(1)
n, m, F = 4096, 1 << 24, 256
t, rx = qpsk(rng, n), cn(rng, m)
d0, k0 = 5_000_000, 37
rx[d0 : d0 + n] += t * np.exp(2j * np.pi * k0 * np.arange(n) / n).astype(np.complex64)
fs = float(n)
f1, f2, step = -128.0, 127.0, 1.0  # the 256 on-grid bins as a CZT grid (whole number of steps)
X._CZTXCORR_FORCE_ROWS = False
for it in range(2):
    t0 = time.perf_counter()
    caf, freqs = X.cztXcorr(t, rx, f1, f2, fs, step, outputCAF=True)
    dt = time.perf_counter() - t0
    pk = np.unravel_index(np.argmax(caf[d0 - 10 : d0 + 10]), (20, F))
    print("%s: cztXcorr(outputCAF=True) call %d: %.2f s  -> %s %s, peak row %d bin %g (%.4f)" % (
        mode, it, dt, caf.shape, caf.dtype, d0 - 10 + pk[0], freqs[pk[1]], caf[d0, pk[1]]), flush=True)
    chk = float(caf[::100003].sum())
    del caf
print("%s: checksum of every 100003rd row: %.9f" % (mode, chk))
# the pieces: plan creation, upload, the launch alone, the download alone
lib = _lib.load()
t0 = time.perf_counter(); d_rx = asarray(rx); t_up = time.perf_counter() - t0
t0 = time.perf_counter(); plan = CAFPlan(t, max_rx_len=m, freqs_norm=(f1 + np.arange(F) * step) / fs); t_plan = time.perf_counter() - t0
kw = dict(surface=True) if mode == "old" else dict(surface_t=True)
res = plan.run(d_rx, rows=False, peak=False, **kw)
_lib.check(lib.caf_stream_sync(None))
t0 = time.perf_counter()
for _ in range(5):
    plan.run(d_rx, rows=False, peak=False, out=res, **kw)
_lib.check(lib.caf_stream_sync(None))
t_launch = (time.perf_counter() - t0) / 5
S = m - n + 1
if mode == "old":
    t0 = time.perf_counter(); h = res.surface.get(); t_down = time.perf_counter() - t0
    t0 = time.perf_counter(); h64 = h[0].astype(np.float64); t_wide = time.perf_counter() - t0
    print("old: upload %.3f s, plan %.3f s, launch %.2f ms, download (17 GB, plain) %.2f s = %.1f GB/s, astype(float64) %.2f s" % (
        t_up, t_plan, t_launch * 1e3, t_down, h.nbytes / t_down / 1e9, t_wide))
else:
    import ctypes as ct
    out = np.empty((S, F), np.float64)
    t0 = time.perf_counter()
    _lib.check(lib.caf_d2h_transposed(out.ctypes.data, 1, ct.c_void_p(res.surface_t.ptr), F, S, 0, S, None))
    t_down = time.perf_counter() - t0
    t0 = time.perf_counter()
    _lib.check(lib.caf_d2h_transposed(out.ctypes.data, 1, ct.c_void_p(res.surface_t.ptr), F, S, 0, S, None))
    t_down2 = time.perf_counter() - t0
    print("new: upload %.3f s, plan %.3f s, launch %.2f ms, transposing download into float64 (17 GB over the link, 34 GB written) "
          "%.2f s first touch, %.2f s into touched pages = %.1f GB/s of device data" % (t_up, t_plan, t_launch * 1e3, t_down, t_down2, 4.0 * S * F / t_down2 / 1e9))
