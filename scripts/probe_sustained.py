"""Is the C2 launch time clock/power-limited?  Times k_caf_persistent (stage timer = HIP events on its stream) launched
back to back against the same launch after idle gaps, and samples sclk / power from rocm-smi while the loop runs."""
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd import CAFPlan, _lib, asarray  # noqa: E402

N, M, F = 4096, 1 << 24, 256
rng = np.random.default_rng(1)
t = qpsk(rng, N)
rx = cn(rng, M)
d_rx = asarray(rx)
bins = np.arange(-F // 2, F // 2)
plan = CAFPlan(t, max_rx_len=M, bins=bins, grid=N)
plan.profile(True)
lib = _lib.load()


def sync():
    _lib.check(lib.caf_stream_sync(None))


def stage_ms():
    p = plan.profile_get()
    ms, n = p["spectral_conj_multiply"]
    return ms, n


res = plan.run(d_rx, surface=True, rows=True, peak=True)
sync()
samples = []
stop = False


def smi():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), o.strip().splitlines()[-1]))
        except Exception as e:  # noqa: BLE001
            samples.append((time.perf_counter(), "smi failed: %r" % e))
        time.sleep(0.25)


th = threading.Thread(target=smi, daemon=True)
th.start()
for surface in (True, False):
    for gap in (0.0, 0.05, 0.5):
        last = stage_ms()
        per = []
        t_begin = time.perf_counter()
        for i in range(40 if gap == 0 else 8):
            plan.run(d_rx, surface=surface, rows=True, peak=True, out=res)
            if gap:
                sync()
                time.sleep(gap)
            if gap or i % 10 == 9:
                sync()
                now = stage_ms()
                per.append((now[0] - last[0]) / max(1, now[1] - last[1]))
                last = now
        sync()
        print("surface=%d gap=%.2fs: kernel ms per launch %s  (wall %.1f s)" % (surface, gap, " ".join("%.2f" % x for x in per),
                                                                          time.perf_counter() - t_begin), flush=True)
stop = True
th.join()
for ts, s in samples[:: max(1, len(samples) // 24)]:
    print("smi", s)
