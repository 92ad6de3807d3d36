"""Diagnosis of the 100 ms settle of cztXcorr's per-delay form with large host arrays (profiles/r04/timing_cztxcorr.log).
One variant per process:  python scripts/diag_stall.py VARIANT [ncalls]
  base     X.cztXcorr, per-delay form forced, per-call wall time
  steps    the same sequence spelled out, every step timed (upload rx / upload cutout / products / CZT / download)
  keep     steps, every host result kept alive (nothing is returned to the OS between calls)
  devin    steps with device-resident inputs (download only)
  noget    steps without the download (device work only, synchronised)
  reuse    steps, download into ONE preallocated host array
  small    steps at 20000 x 101 x 401 (the case that never stalled), for the log
"""
import ctypes as ct
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn  # noqa: E402
import pydsproutines_amd.xcorrRoutines as X  # noqa: E402
from pydsproutines_amd import _lib  # noqa: E402
from pydsproutines_amd.cupyExtensions import multiplySlidesNormalised  # noqa: E402
from pydsproutines_amd.devarray import asarray, pool_stats  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "base"
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(1)
if variant == "small":
    n, m, nsh, span, step = 20_000, 40_000, 101, 50.0, 0.25
else:
    n, m, nsh, span, step = 100_000, 120_000, 201, 100.0, 0.1
cut, rx = cn(rng, n), cn(rng, m)
sh = np.arange(5000, 5000 + nsh)
lib = _lib.load()


def sync():
    _lib.check(lib.caf_stream_sync(None), "sync")


if variant == "base":
    X._CZTXCORR_FORCE_ROWS = True
    ts = []
    for _ in range(ncalls):
        t0 = time.perf_counter()
        X.cztXcorr(cut, rx, -span, span, 1e5, step, False, sh)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("base  per-call ms:", " ".join("%.1f" % t for t in ts), "| pool", pool_stats(), flush=True)
    sys.exit(0)

czt = X._czt_object(n, -span, span, step, 1e5)
keep = []
d_rx0, d_cut0 = asarray(X._c64(rx)), asarray(X._c64(cut).conj())
host_out = np.empty((nsh, czt.k), np.complex64)
print("variant %s: nfft %d, k %d, rows %d" % (variant, czt.nfft, czt.k, nsh), flush=True)
for it in range(ncalls):
    t = [time.perf_counter()]
    d_rx = d_rx0 if variant == "devin" else asarray(X._c64(rx))
    t.append(time.perf_counter())
    d_cut = d_cut0 if variant == "devin" else asarray(X._c64(cut).conj())
    t.append(time.perf_counter())
    d_p = multiplySlidesNormalised(d_cut, d_rx, 5000, nsh)
    sync()
    t.append(time.perf_counter())
    d_spec = czt.runMany(d_p)
    sync()
    t.append(time.perf_counter())
    if variant == "noget":
        spec = None
    elif variant == "reuse":
        _lib.check(lib.caf_d2h(host_out.ctypes.data, ct.c_void_p(d_spec.ptr), host_out.nbytes, None), "caf_d2h")
        spec = host_out
    else:
        spec = d_spec.get()
    t.append(time.perf_counter())
    if variant == "keep":
        keep.append(spec)
    del spec
    d = np.diff(t) * 1e3
    print("  call %d: h2d rx %.2f  h2d cutout %.2f  products %.2f  czt %.2f  d2h %.2f  | total %.2f ms" % ((it,) + tuple(d) + (d.sum(),)), flush=True)
print("pool", pool_stats(), flush=True)
