"""Debug: persistent vs fused engine, 2 templates x 128 bins, two trials through the same plans; prints
the delays whose surface rows / row results differ (the two engines do identical arithmetic)."""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk
from pydsproutines_amd import CAFPlan, asarray

rng = np.random.default_rng(77)
n, m, F = 512, 40000, 128
b = np.arange(-F // 2, F // 2)
tm = np.stack([qpsk(rng, n) for _ in range(2)])
pa = CAFPlan(tm, max_rx_len=m, bins=b, grid=n, engine="persistent")
pb = CAFPlan(tm, max_rx_len=m, bins=b, grid=n, engine="fused")
print("step", pa.step, "tiles/blk", -(-pa.step // 64))
for trial in range(3):
    rx = cn(rng, m)
    rx[2000 + 7000 * trial : 2000 + 7000 * trial + n] += 2 * tm[trial % 2]
    d = asarray(rx)
    ra, rb = pa.run(d, surface=True), pb.run(d, surface=True)
    sa, sb = ra.surface.get(), rb.surface.get()
    for t in range(2):
        bad = np.where(np.any(sa[t] != sb[t], axis=1))[0]
        badr = np.where((ra.row_max.get()[t] != rb.row_max.get()[t]) | (ra.row_arg.get()[t] != rb.row_arg.get()[t]))[0]
        print("trial", trial, "template", t, "surface rows differing:", bad.size, bad[:12], "tiles", np.unique(bad // 64)[:12] if bad.size else "",
              "| row results differing:", badr.size, badr[:8], flush=True)
        if bad.size:
            r = bad[0]
            cols = np.where(sa[t][r] != sb[t][r])[0]
            print("   first bad row", r, "cols", cols[:10], "n", cols.size, "got", sa[t][r][cols[:4]], "want", sb[t][r][cols[:4]])
    print("  peaks", ra.peak_delay.get(), rb.peak_delay.get(), ra.peak_freq.get(), rb.peak_freq.get())
