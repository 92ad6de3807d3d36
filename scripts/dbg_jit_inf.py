import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import cn
from test_gpu_perdelay import _perdelay
rng = np.random.default_rng(12)
n, m = 1200, 30_000
rx, cut = cn(rng, m), cn(rng, n)
for num, step, start in ((4000, 5, 3), (101, 5, 3), (4000, 1, 3), (4000, 5, 0), (1334 * 3, 5, 3), (300, 5, 3), (1000, 5, 3), (3000, 5, 3)):
    res = {}
    for jit in ("1", "0"):
        os.environ["CAF_JIT"] = jit
        q = _perdelay(cut, rx, start, step, num, caf=True)[0]
        res[jit] = q
    bad = ~np.isfinite(res["1"])
    print("num %5d step %d start %d: jit finite %5d / %5d, max diff on finite %.2e, first bad rows %s" % (
        num, step, start, (~bad).sum(), num, np.max(np.abs(res["1"][~bad] - res["0"][~bad])) if (~bad).any() else -1, np.nonzero(bad)[0][:8]))
