"""Seeded random shapes through the three inverse-transform engines: ragged delay tiles, frequency counts
that are not multiples of the tile-role step, several templates, on-grid and explicit-frequency hypotheses,
sub-ranges of delays, several batches.  Checked per case:
  * persistent == fused bit for bit (same arithmetic, different launch structure);
  * rocfft (independent FFT) agrees within float tolerance;
  * sampled rows against the oracle (the reference's per-delay algorithm), surface within 1e-4 of its maximum;
  * row results are exactly the max / first argmax of the surface the GPU wrote; the planted peak is found."""

import numpy as np
import pytest

import oracle as O
from conftest import cn, qpsk

pytestmark = pytest.mark.gpu

import os

CASES = list(range(int(os.environ.get("CAF_FUZZ_CASES", "28"))))  # CAF_FUZZ_CASES=400 for a soak run


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([64, 100, 256, 500, 1024, 2048]))
    t = int(rng.choice([1, 1, 2, 3, 5]))
    f = int(rng.choice([1, 3, 16, 31, 32, 33, 64, 70, 96, 130]))
    m = int(rng.integers(n + 200, 60000))
    table = bool(rng.integers(0, 2)) and f > 1
    grid = 1 << int(np.ceil(np.log2(n)))
    if table:
        freqs = np.sort(rng.uniform(-0.02, 0.02, f))
    else:
        lo = -(f // 2)
        bins = np.arange(lo, lo + f)
    tm = np.stack([qpsk(rng, n) for _ in range(t)])
    # composite template (GroupXcorr semantics): 1-3 groups with gaps inside the span; zeros in the gaps and the
    # energy normalisation over the support only
    gs = gl = None
    if n >= 100 and rng.integers(0, 3) == 0:
        G = int(rng.integers(1, 4))
        cuts = np.sort(rng.choice(np.arange(1, n), 2 * G - 1, replace=False))
        edges = np.concatenate(([0], cuts, [n]))
        gs = edges[0::2][:G].astype(np.int32)
        gl = (edges[1::2][:G] - gs).astype(np.int32)
        mask = np.zeros(n, bool)
        for a, l in zip(gs, gl):
            mask[a : a + l] = True
        tm = (tm * mask).astype(np.complex64)
    rx = cn(rng, m)
    s_total = m - n + 1
    truth = []
    for i in range(t):
        d = int(rng.integers(0, s_total))
        j = int(rng.integers(0, f))
        nu = freqs[j] if table else bins[j] / grid
        rx[d : d + n] += (3 * tm[i] * np.exp(2j * np.pi * nu * np.arange(n))).astype(np.complex64)
        truth.append((d, j))
    kw = dict(freqs_norm=freqs) if table else dict(bins=bins, grid=grid)
    nb = int(rng.choice([0, 0, 1, 2]))
    sub = bool(rng.integers(0, 2))
    lo_s = int(rng.integers(0, max(1, s_total // 3))) if sub else 0
    cnt = int(rng.integers(1, s_total - lo_s + 1)) if sub else s_total
    if gs is not None:
        kw.update(group_starts=gs, group_lens=gl)
    conj_by_caller = bool(rng.integers(0, 4) == 0)  # autoConj=False: the caller hands over conj(template)
    lb = int(rng.integers(max(9, int(np.ceil(np.log2(2 * n)))), 17))  # block size of the rocfft engine
    return dict(n=n, t=t, f=f, m=m, tm=tm, rx=rx, kw=kw, nb=nb, lo=lo_s, cnt=cnt, truth=truth, table=table,
                nu=(freqs if table else bins / grid), gs=gs, gl=gl, conj_by_caller=conj_by_caller, lb=lb)


@pytest.mark.parametrize("seed", CASES)
def test_random_shapes_all_engines(seed):
    from pydsproutines_amd import CAFPlan, asarray

    c = _case(seed)
    d_rx = asarray(c["rx"])
    out = {}
    for engine in ("persistent", "fused", "rocfft"):
        tmpl = c["tm"].conj() if c["conj_by_caller"] else c["tm"]
        plan = CAFPlan(tmpl, max_rx_len=c["m"], engine=engine, blocks_per_batch=c["nb"],
                       autoConj=not c["conj_by_caller"], log2_block=c["lb"] if engine == "rocfft" else 0, **c["kw"])
        assert plan.engine_used == engine
        r = plan.run(d_rx, shift_start=c["lo"], num_shifts=c["cnt"], surface=True)
        out[engine] = (r.surface.get(), r.row_max.get(), r.row_arg.get(), r.peak_val.get(), r.peak_delay.get(),
                       r.peak_freq.get())
        # the same call without a surface (the persistent engine then keeps running maxima instead of tiles
        # where the hypothesis groups allow it): identical per-delay results and peaks
        r2 = plan.run(d_rx, shift_start=c["lo"], num_shifts=c["cnt"], surface=False, rows=True, peak=True)
        surf = out[engine][0]
        for name, a, b in zip(("row_max", "row_arg", "peak_val", "peak_delay", "peak_freq"), out[engine][1:],
                              (r2.row_max.get(), r2.row_arg.get(), r2.peak_val.get(), r2.peak_delay.get(),
                               r2.peak_freq.get())):
            if name == "row_arg":
                # The running maxima compare |y|^2 before the common normalisation factor is applied, the surface
                # path after it: two hypotheses whose normalised float32 values round to the same number are a tie
                # there (first index wins) but not here (the larger raw value wins).  Any index the no-surface run
                # reports must therefore hold the row maximum, and differ from the surface run only on such ties.
                ti, si = np.nonzero(a != b)
                # (rare: a handful of rows per case at most, and only with near-duplicate explicit frequencies)
                assert ti.size <= max(16, a.size // 1000), "%d differing per-delay arguments" % ti.size
                np.testing.assert_array_equal(surf[ti, si, b[ti, si]], out[engine][1][ti, si])
                assert np.all(b[ti, si] > a[ti, si])
            elif name == "peak_freq":
                np.testing.assert_array_equal(surf[np.arange(a.size), out[engine][4] - c["lo"], b], out[engine][3])
            else:
                np.testing.assert_array_equal(a, b)
        r3 = plan.run(d_rx, shift_start=c["lo"], num_shifts=c["cnt"], surface=False, rows=False, peak=True)
        np.testing.assert_array_equal(r3.peak_delay.get(), out[engine][4])
        np.testing.assert_array_equal(r3.peak_val.get(), out[engine][3])
        plan.close()
    sp, rmp, rap, pvp, pdp, pfp = out["persistent"]
    for a, b in zip(out["persistent"], out["fused"]):
        np.testing.assert_array_equal(a, b)
    sr, rmr, rar, pvr, pdr, pfr = out["rocfft"]
    scale = float(np.nanmax(sp))
    # The correlation is computed on whole overlap-save blocks, so its absolute float32 error scales with the
    # block's energy, not with the energy under the template: a composite template with only a few samples of
    # support (QF^2 ~ 1 on anything) amplifies it -- the tolerance between the two FFTs widens accordingly.
    support = c["n"] if c["gl"] is None else int(np.sum(c["gl"]))
    assert np.nanmax(np.abs(sp - sr)) <= 2e-5 * max(scale, 1e-3) * max(1.0, 64.0 / support)
    # self-consistency of the persistent engine's outputs
    assert sp.shape == (c["t"], c["cnt"], c["f"])
    np.testing.assert_array_equal(rmp, sp.max(axis=2))
    np.testing.assert_array_equal(rap, np.argmax(sp, axis=2))
    for i in range(c["t"]):
        j = int(np.argmax(rmp[i]))
        assert pvp[i] == rmp[i][j] and pdp[i] == c["lo"] + j and pfp[i] == rap[i][j]
        d, fj = c["truth"][i]
        support = c["n"] if c["gl"] is None else int(np.sum(c["gl"]))
        # (a composite template with a handful of samples scores QF^2 ~ 1 on noise alone: no planted-peak check)
        if c["lo"] <= d < c["lo"] + c["cnt"] and support >= 32:
            # (explicit frequency lists may hold near-duplicates, which noise can lift above the planted one)
            assert int(pdp[i]) == d, "planted delay of template %d" % i
            # (a sanity check of the test's own planting, not of the engines: 1 of 3000 seeds has a neighbouring
            # frequency of a short composite template 3 % above the planted one)
            assert sp[i][d - c["lo"]][fj] >= 0.9 * pvp[i], "planted frequency of template %d" % i
    # sampled rows against the oracle
    rng = np.random.default_rng(seed)
    rows = np.unique(np.concatenate((rng.integers(0, c["cnt"], 24), [0, c["cnt"] - 1])))
    refs = []
    for i in range(c["t"]):
        ref = _oracle_rows(c["tm"][i], c["rx"], c["nu"], c["lo"] + rows, c["gs"], c["gl"])
        refs.append(ref)
        tol = 1e-4 * max(float(ref.max()), float(scale))
        assert np.max(np.abs(sp[i][rows] - ref)) <= tol
    # templates with at most 64 non-zero samples: the direct engine (what AUTO picks for small composite templates) holds
    # the oracle at a tolerance that does not depend on the support, and the overlap-save engines at theirs
    if np.count_nonzero(np.any(c["tm"] != 0, axis=0)) <= 64:
        tmpl = c["tm"].conj() if c["conj_by_caller"] else c["tm"]
        plan = CAFPlan(tmpl, max_rx_len=c["m"], engine="direct", autoConj=not c["conj_by_caller"], **c["kw"])
        assert plan.engine_used == "direct"
        r = plan.run(d_rx, shift_start=c["lo"], num_shifts=c["cnt"], surface=True)
        sd = r.surface.get()
        for i in range(c["t"]):
            assert np.max(np.abs(sd[i][rows] - refs[i])) <= 4e-6 * max(float(refs[i].max()), 1e-3)
        assert np.nanmax(np.abs(sd - sp)) <= 2e-5 * max(scale, 1e-3) * max(1.0, 64.0 / support)
        np.testing.assert_array_equal(r.row_max.get(), sd.max(axis=2))
        np.testing.assert_array_equal(r.row_arg.get(), np.argmax(sd, axis=2))
        r2 = plan.run(d_rx, shift_start=c["lo"], num_shifts=c["cnt"], surface=False)
        np.testing.assert_array_equal(r2.row_max.get(), r.row_max.get())
        np.testing.assert_array_equal(r2.peak_delay.get(), r.peak_delay.get())
        np.testing.assert_array_equal(r2.peak_freq.get(), r.peak_freq.get())
        plan.close()


CHAINED_CASES = list(range(max(10, int(os.environ.get("CAF_FUZZ_CASES", "28")) // 2)))


def _chained_case(seed):
    """Templates of 8193 .. 262144 samples: 32768-point blocks (two chained transforms; one, two or all four quarters of the
    upper half valid) and 65536-point blocks (folded: tiles of every second delay; beyond 32768 samples the template in
    partitions of 32768 samples against the spectra of the blocks that follow)."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([8193, 9000, 12288, 12289, 16383, 16384, 16385, 20000, 24576, 32767, 32768,
                        32769, 40000, 65535, 65536, 65537, 90000, 98304, 131072, 150000, 262144]))  # (from 32769: 2 .. 8 template partitions)
    t = int(rng.choice([1, 1, 2, 3]))
    f = int(rng.choice([1, 2, 5, 31, 33, 64, 70]))
    step = (32768 if n <= 16384 else 65536) - n + 1 if n <= 16384 else 32768
    nblk = float(rng.choice([0.3, 1.0, 1.7, 2.0, 3.2]))
    s_total = max(1, int(nblk * step) + int(rng.integers(-70, 70)))
    m = n + s_total - 1
    table = bool(rng.integers(0, 2)) and f > 1
    grid = 16384
    if table:
        freqs = np.sort(rng.uniform(-2e-3, 2e-3, f))
    else:
        lo = -(f // 2)
        bins = np.arange(lo, lo + f)
    tm = np.stack([qpsk(rng, n) for _ in range(t)])
    gs = gl = None
    if rng.integers(0, 4) == 0:  # composite template: two groups with a gap
        a, b = sorted(int(v) for v in rng.choice(np.arange(1000, n - 1000), 2, replace=False))
        gs, gl = np.array([0, b], np.int32), np.array([a, n - b], np.int32)
        mask = np.zeros(n, bool)
        mask[:a] = True
        mask[b:] = True
        tm = (tm * mask).astype(np.complex64)
    rx = cn(rng, m)
    truth = []
    for i in range(t):
        d = int(rng.integers(0, s_total))
        j = int(rng.integers(0, f))
        nu = freqs[j] if table else bins[j] / grid
        rx[d : d + n] += (tm[i] * np.exp(2j * np.pi * nu * np.arange(n))).astype(np.complex64)
        truth.append((d, j))
    kw = dict(freqs_norm=freqs) if table else dict(bins=bins, grid=grid)
    if gs is not None:
        kw.update(group_starts=gs, group_lens=gl)
    sub = bool(rng.integers(0, 2))
    lo_s = int(rng.integers(0, max(1, s_total // 2))) if sub else 0
    cnt = int(rng.integers(1, s_total - lo_s + 1)) if sub else s_total
    return dict(n=n, t=t, f=f, m=m, tm=tm, rx=rx, kw=kw, lo=lo_s, cnt=cnt, truth=truth, nu=(freqs if table else bins / grid),
                gs=gs, gl=gl, nb=int(rng.choice([0, 0, 1])))


@pytest.mark.parametrize("seed", CHAINED_CASES)
def test_random_shapes_chained_roles(seed):
    """The chained roles of the persistent engine against the rocfft engine and the oracle, and against themselves: the rows
    and the peak are those of the surface written; without the surface the per-delay results are the same bits."""
    from pydsproutines_amd import CAFPlan, asarray

    c = _chained_case(seed)
    d_rx = asarray(c["rx"])
    plan = CAFPlan(c["tm"], max_rx_len=c["m"], engine="persistent", blocks_per_batch=c["nb"], **c["kw"])
    assert plan.engine_used == "persistent" and plan.block == (32768 if c["n"] <= 16384 else 65536)
    r = plan.run(d_rx, shift_start=c["lo"], num_shifts=c["cnt"], surface=True)
    sp, rmp, rap = r.surface.get(), r.row_max.get(), r.row_arg.get()
    pvp, pdp, pfp = r.peak_val.get(), r.peak_delay.get(), r.peak_freq.get()
    assert sp.shape == (c["t"], c["cnt"], c["f"]) and not np.any(np.isnan(sp))
    np.testing.assert_array_equal(rmp, sp.max(axis=2))
    np.testing.assert_array_equal(rap, np.argmax(sp, axis=2))
    for i in range(c["t"]):
        j = int(np.argmax(rmp[i]))
        assert pvp[i] == rmp[i][j] and pdp[i] == c["lo"] + j and pfp[i] == rap[i][j]
        d, fj = c["truth"][i]
        if c["lo"] <= d < c["lo"] + c["cnt"]:
            assert int(pdp[i]) == d, "planted delay of template %d" % i
    r2 = plan.run(d_rx, shift_start=c["lo"], num_shifts=c["cnt"], surface=False, rows=True, peak=True)
    np.testing.assert_array_equal(r2.row_max.get(), rmp)
    np.testing.assert_array_equal(r2.row_arg.get(), rap)
    np.testing.assert_array_equal(r2.peak_delay.get(), pdp)
    np.testing.assert_array_equal(r2.peak_freq.get(), pfp)
    r3 = plan.run(d_rx, shift_start=c["lo"], num_shifts=c["cnt"], surface=False, rows=False, peak=True)
    np.testing.assert_array_equal(r3.peak_delay.get(), pdp)
    np.testing.assert_array_equal(r3.peak_val.get(), pvp)
    plan.close()
    q = CAFPlan(c["tm"], max_rx_len=c["m"], engine="rocfft", **c["kw"])
    sr = q.run(d_rx, shift_start=c["lo"], num_shifts=c["cnt"], surface=True).surface.get()
    q.close()
    scale = float(np.nanmax(sp))
    assert np.nanmax(np.abs(sp - sr)) <= 2e-5 * max(scale, 1e-3)
    rng = np.random.default_rng(seed)
    rows = np.unique(np.concatenate((rng.integers(0, c["cnt"], 6), [0, c["cnt"] - 1])))
    for i in range(c["t"]):
        ref = _oracle_rows(c["tm"][i], c["rx"], c["nu"], c["lo"] + rows, c["gs"], c["gl"])
        assert np.max(np.abs(sp[i][rows] - ref)) <= 1e-4 * max(float(ref.max()), scale)


def _oracle_rows(tmpl, rx, nu, shifts, gs=None, gl=None):
    """QF^2 at the given delays and normalised frequencies: the reference's per-delay definition
    (xcorrRoutines.py:511-566) with an explicit DFT row per frequency, float64; with groups, the rx energy is
    taken over the support of the composite template only (GroupXcorr, xcorrRoutines.py:917-954)."""
    n = tmpl.size
    k = np.arange(n)
    sup = np.ones(n, bool)
    if gs is not None:
        sup[:] = False
        for a, l in zip(gs, gl):
            sup[a : a + l] = True
    e_t = float(np.sum(np.abs(tmpl.astype(np.complex128)) ** 2))
    steer = np.exp(-2j * np.pi * np.outer(np.asarray(nu, dtype=np.float64), k))
    out = np.empty((shifts.size, steer.shape[0]))
    for a, s in enumerate(shifts):
        seg = rx[s : s + n].astype(np.complex128)
        p = seg * np.conj(tmpl.astype(np.complex128))
        out[a] = np.abs(steer @ p) ** 2 / (e_t * float(np.sum(np.abs(seg[sup]) ** 2)))
    return out


def test_oracle_rows_helper_matches_the_oracle(golden):
    """The explicit-DFT row helper above is the same quantity as oracle.caf_bins on the on-grid case."""
    g = golden("c2_mini")
    ref = O.caf_bins(g["template"], g["rx"], g["bins"], g["shifts"][:20])
    got = _oracle_rows(g["template"], g["rx"], g["bins"] / g["template"].size, g["shifts"][:20])
    np.testing.assert_allclose(got, ref, atol=1e-6)  # the oracle keeps the reference's complex64 arithmetic
