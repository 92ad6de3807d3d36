"""Where the wall time of one small fastXcorr call goes (cProfile, cumulative)."""
import cProfile
import pstats
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import cn  # noqa: E402
from pydsproutines_amd.xcorrRoutines import fastXcorr  # noqa: E402

rng = np.random.default_rng(0)
rx = cn(rng, 65536)
cut = rx[1000:2024].copy()
for _ in range(3):
    fastXcorr(cut, rx)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    fastXcorr(cut, rx)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
