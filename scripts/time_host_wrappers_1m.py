"""Host-signature wrappers asked for about a million delays: where the wall clock goes beside the device work (cProfile, 3 calls each)."""
import cProfile
import pstats
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn, qpsk  # noqa: E402
from pydsproutines_amd.xcorrRoutines import CyIppXcorrFFT, GroupXcorr, GroupXcorrFFT, cztXcorr, fastXcorr  # noqa: E402

rng = np.random.default_rng(0)
rx = cn(rng, 1 << 20)
fs = 1000.0
y = qpsk(rng, 4096)
sh = np.arange(0, (1 << 20) - 4096)


def prof(name, fn):
    fn()
    fn()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        fn()
    pr.disable()
    st = pstats.Stats(pr)
    tot = st.total_tt / 3
    print("== %s: %.2f ms per call" % (name, tot * 1e3), flush=True)
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:6]  # by tottime
    for (f, line, fname), (cc, nc, tt, ct, _) in rows:
        print("      %6.2f ms own  %s:%d %s" % (tt / 3 * 1e3, f.split("/")[-1], line, fname))


g = GroupXcorr(y, np.array([0, 2048]), np.array([1024, 1024]), np.arange(-8, 8) * fs / 4096, fs)
prof("GroupXcorr.xcorr 2 groups x 1024, 16 freqs, 1.04 M shifts", lambda: g.xcorr(rx, sh))
gf = GroupXcorrFFT(np.stack([y[:256], y[1000:1256]]), np.array([0, 1000]), fs, fftlen=256)
prof("GroupXcorrFFT.xcorr 2 groups x 256, fftlen 256, 1.04 M shifts", lambda: gf.xcorr(rx, sh))
prof("cztXcorr 4096 cutout, 17 bins, 1.04 M shifts (flattened)", lambda: cztXcorr(y, rx, -1.0, 1.0, fs, 0.125, shifts=sh))
cy = CyIppXcorrFFT(y[:1024], 4, True)
prof("CyIppXcorrFFT(1024).xcorr 1 M delays", lambda: cy.xcorr(rx, 0, 1_000_000, 1))
prof("fastXcorr(freqsearch=True) 1024 cutout, 1 M delays", lambda: fastXcorr(y[:1024], rx, freqsearch=True, shifts=np.arange(1_000_000)))
