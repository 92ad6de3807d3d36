"""The reference's benchmark call sequences at THEIR sizes (tests/test_gpu_reference_benchmarks.py runs them reduced):
benchmark_groupXcorrs.py (10^6 QPSK symbols, 100 groups of 5000, 201 CZT bins, 41 shifts), benchmark_czts.py (ten rows of
10000, 2001 bins), benchmark_cupyTemplateCrossCorrelator.py (8 cutouts of 1000 in 200000 samples... scaled to its own
defaults).  Wall clock per call, after one warm-up call."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from pydsproutines_amd import _lib, asarray  # noqa: E402
from pydsproutines_amd.signalCreationRoutines import randnoise, randPSKsyms  # noqa: E402
from pydsproutines_amd.spectralRoutines import CZTCached, CZTCachedGPU, pbIppCZT32fc  # noqa: E402
from pydsproutines_amd.xcorrRoutines import GroupXcorrCZT, TemplateCrossCorrelator, pbIppGroupXcorrCZT  # noqa: E402


def sync():
    _lib.check(_lib.load().caf_stream_sync(None))


def timed(name, f, reps=3):
    f()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = f()
    sync()
    print("%-64s %9.2f ms per call" % (name, (time.perf_counter() - t0) / reps * 1e3), flush=True)
    return r


# ---- benchmark_groupXcorrs.py:19-72
np.random.seed(5)
x, _ = randPSKsyms(1_000_000, 4, dtype=np.complex64)
f1, f2, fstep, fs = -100.0, 100.0, 1.0, 10000
firstGroupStart, groupLength = 100, 5000
groupStarts = np.arange(firstGroupStart, x.size, groupLength * 2, dtype=np.int32)
t0 = time.perf_counter()
gxc = GroupXcorrCZT(x, groupStarts, np.zeros(groupStarts.size, dtype=np.int32) + groupLength, f1, f2, fstep, fs)
print("%-64s %9.2f ms" % ("GroupXcorrCZT(...) construction, %d groups of %d" % (groupStarts.size, groupLength), (time.perf_counter() - t0) * 1e3))
shifts = np.arange(firstGroupStart - 20, firstGroupStart - 20 + 41)
res, _ = timed("GroupXcorrCZT.xcorr(x, 41 shifts) -> (41, 201)", lambda: gxc.xcorr(x, shifts))
assert np.unravel_index(np.argmax(res), res.shape) == (20, 100)
pb = pbIppGroupXcorrCZT(groupLength, f1, f2, fstep, fs, 4)
for gs in groupStarts:
    pb.addGroup(gs - firstGroupStart, x[gs : gs + groupLength])
pres = timed("pbIppGroupXcorrCZT(4 threads).xcorr(x, 80, 1, 41)", lambda: pb.xcorr(x, firstGroupStart - 20, 1, 41))
assert np.unravel_index(np.argmax(pres), pres.shape) == (20, 100)

# ---- benchmark_czts.py:20-82
np.random.seed(7)
length = 10000
xs = np.vstack([randnoise(length, 1, 1, 10).astype(np.complex64) for _ in range(10)])
d_cztobj = CZTCachedGPU(length, -1000.0, 1000.0, 1.0, length)
d_xs = asarray(xs)
timed("CZTCachedGPU.runMany(10 x 10000) -> (10, 2001)", lambda: d_cztobj.runMany(d_xs))
cz = CZTCached(length, -1000.0, 1000.0, 1.0, length, True)
timed("CZTCached.runMany (host arrays in / out)", lambda: cz.runMany(xs))
pbczt = pbIppCZT32fc(length, -1000.0, 1000.0, 1.0, float(length))
timed("pbIppCZT32fc.runMany (host arrays in / out)", lambda: pbczt.runMany(xs))

# ---- benchmark_cupyTemplateCrossCorrelator.py:19-37 (its defaults: 10^6 samples, 100 cutouts of 1000)
np.random.seed(6)
length, cutoutlen, numCutouts = 1_000_000, 1000, 100
y, _ = randPSKsyms(length, 4, dtype=np.complex64)
cutouts = np.stack([y[5000 + 9000 * i : 5000 + 9000 * i + cutoutlen] for i in range(numCutouts)])
dx, dc = asarray(y), asarray(cutouts)
tcc = TemplateCrossCorrelator(dc, dx.size)
out, ti = timed("TemplateCrossCorrelator(100 x 1000, 10^6).correlate(returnMax=True)", lambda: tcc.correlate(dx, returnMax=True))
o = out.get()
assert abs(o[5000] - 1.0) < 1e-5 and ti.get()[5000] == 0
