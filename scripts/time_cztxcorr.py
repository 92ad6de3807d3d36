"""cztXcorr (xcorrRoutines.py:413-457) as a fine frequency search over a few delays of a long cutout -- the per-delay form
(product rows, one batched CZT) against the hypothesis engine -- and over all delays of a short one.
usage: python scripts/time_cztxcorr.py [rule|rows|engine ...]   (one form per process: the forms leave different pools behind)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn  # noqa: E402
import pydsproutines_amd.xcorrRoutines as X  # noqa: E402

rng = np.random.default_rng(1)
for n, m, nsh, span, step in ((100_000, 120_000, 201, 100.0, 0.1), (20_000, 40_000, 101, 50.0, 0.25), (8000, 40_000, 41, 20.0, 0.1), (4096, 65536, None, 20.0, 0.5)):
    cut = cn(rng, n)
    rx = cn(rng, m)
    sh = None if nsh is None else np.arange(5000, 5000 + nsh)
    for force in ({"rule": None, "rows": True, "engine": False}[a] for a in (sys.argv[1:] or ["rule", "rows", "engine"])):
        X._CZTXCORR_FORCE_ROWS = force
        X.cztXcorr(cut, rx, -span, span, 1e5, step, False, sh)
        t0 = time.perf_counter()
        for _ in range(5):
            X.cztXcorr(cut, rx, -span, span, 1e5, step, False, sh)
        dt = (time.perf_counter() - t0) / 5
        print("cztXcorr cutout %6d, %s shifts, %4d bins, %-12s %8.2f ms per call (host arrays in / out)" % (
            n, "all" if nsh is None else str(nsh), int(2 * span / step + 1), {None: "rule", True: "per-delay", False: "engine"}[force], dt * 1e3), flush=True)
