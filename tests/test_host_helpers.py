"""Host-side helpers of the reference-signature layer that need no GPU."""
import numpy as np


def _runs_reference(shifts):
    """The element-by-element definition: maximal arithmetic progressions, greedily, a repeated index as a run of one."""
    s = np.asarray(shifts, dtype=np.int64).reshape(-1)
    out, i, n = [], 0, s.size
    while i < n:
        if i + 1 >= n:
            out.append((i, int(s[i]), 1, 1))
            break
        step = int(s[i + 1] - s[i])
        j = i + 1
        while j + 1 < n and int(s[j + 1] - s[j]) == step:
            j += 1
        if step == 0:
            step = 1
            j = i
        out.append((i, int(s[i]), step, j - i + 1))
        i = j + 1
    return out


def test_runs_of_a_shift_list():
    """xcorrRoutines._runs (how fastXcorr(freqsearch=True) / CyIppXcorrFFT-style calls cut an arbitrary shift list into
    start / step / count runs for the per-delay kernels): equal to the definition on random lists, and one run for a million
    consecutive delays without a Python loop over them."""
    from pydsproutines_amd.xcorrRoutines import _runs

    rng = np.random.default_rng(0)
    for _ in range(2000):
        n = int(rng.integers(0, 40))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            a = rng.integers(0, 6, n)
        elif kind == 1:
            a = np.cumsum(rng.integers(-2, 3, n))
        elif kind == 2:
            a = np.arange(n) * int(rng.integers(-3, 4)) + 5
        else:
            a = np.concatenate([np.arange(rng.integers(0, 10)), np.arange(rng.integers(0, 10)) * 2 + 50, rng.integers(0, 4, rng.integers(0, 6))])
        assert _runs(a) == _runs_reference(a), a
    assert _runs(np.arange(1_000_000)) == [(0, 0, 1, 1_000_000)]
    assert _runs(np.arange(10, 0, -3)) == [(0, 10, -3, 4)]
