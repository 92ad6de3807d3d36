"""Debug driver of the persistent engine: c2_mini-sized plan, queue dump (CAF_PERSIST_DEBUG=1)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oracle as O
from pydsproutines_amd import CAFPlan, asarray

g = np.load("tests/golden/c2_mini.npz")
t, rx, bins, sh = g["template"], g["rx"], g["bins"], g["shifts"]
for engine in sys.argv[1:] or ["persistent"]:
    plan = CAFPlan(t, max_rx_len=rx.size, bins=bins, grid=t.size, engine=engine)
    print("engine", plan.engine_used, "block", plan.block, flush=True)
    t0 = time.time()
    res = plan.run(asarray(rx), surface=True)
    surf = res.surface.get()[0]
    print("ran in %.3f s" % (time.time() - t0), "max err", float(np.max(np.abs(surf[sh] - g["caf"]))),
          "peak", int(res.peak_delay.get()[0]), int(bins[res.peak_freq.get()[0]]), "expect", int(g["d0"][0]), int(g["k0"][0]), flush=True)
