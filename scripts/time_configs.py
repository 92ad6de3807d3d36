"""Wall-clock of the non-benchmark BASELINE configs per engine (C3: 64 templates, no frequency scan;
C4 per-GPU share: 64 templates x 512 bins), 2^24-sample rx, inputs resident.  Prints one line per run."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd import CAFPlan, _lib, asarray  # noqa: E402

N, M = 4096, 1 << 24
rng = np.random.default_rng(5)
tm = np.stack([qpsk(rng, N) for _ in range(64)])
rx = cn(rng, M)
rx[1_000_000 : 1_000_000 + N] += tm[7]
d_rx = asarray(rx)


def sync():
    _lib.check(_lib.load().caf_stream_sync(None))


cases = [("C3 64 templates x 1 bin, rows+peak", dict(bins=[0]), dict(rows=True, peak=True)),
         ("C3 64 templates x 1 bin, peak only", dict(bins=[0]), dict(rows=False, peak=True)),
         ("C4/GPU 64 templates x 512 bins, peak only", dict(bins=np.arange(-256, 256)), dict(rows=False, peak=True))]
for name, pk, rk in cases:
    for engine in [a for a in sys.argv[1:] if a != "tcc"] or ["persistent", "fused", "rocfft"]:
        try:
            plan = CAFPlan(tm, max_rx_len=M, grid=N, engine=engine, **pk)
        except ValueError as e:
            print(name, engine, "n/a:", e)
            continue
        res = plan.run(d_rx, **rk)
        sync()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            res = plan.run(d_rx, out=res, **rk)
        sync()
        dt = (time.perf_counter() - t0) / reps
        hyps = 64 * len(pk["bins"])
        print("%-44s %-10s %8.1f ms  %7.1f Mdelay-hypotheses/s/1e3  block=%d batch=%d peak7=%d" % (
            name, plan.engine_used, dt * 1e3, hyps * (M - N + 1) / dt / 1e9, plan.block, plan.blocks_per_batch,
            int(res.peak_delay.get()[7])), flush=True)
        plan.close()

# C3 through the reference's own class: TemplateCrossCorrelator.correlate(returnMax=True)
if not sys.argv[1:] or "tcc" in sys.argv[1:]:
    from pydsproutines_amd.xcorrRoutines import TemplateCrossCorrelator  # noqa: E402

    d_tm = asarray(tm)
    for fast in (True, False):
        tcc = TemplateCrossCorrelator(d_tm, M, fastMax=fast)
        qf, ti = tcc.correlate(d_rx, returnMax=True)
        sync()
        t0 = time.perf_counter()
        for _ in range(2):
            qf, ti = tcc.correlate(d_rx, returnMax=True)
        sync()
        dt = (time.perf_counter() - t0) / 2
        j = int(np.argmax(qf.get()))
        print("C3 TemplateCrossCorrelator.correlate(returnMax=True) fastMax=%-5s %8.1f ms  peak at %d template %d" % (
            fast, dt * 1e3, j, int(ti.get()[j])), flush=True)
        del tcc
