"""The hand-off protocol of the one-launch engine (k_caf_persistent, csrc/caf_fused.hip) under every residency and
role split it can be started with: fewer workgroups than CUs, more than can be resident, no tile-first workgroups,
nearly all tile-first; two plans running at once on two streams.  Each case must reproduce the two-launch fused
engine BIT FOR BIT (same arithmetic, different scheduling), terminate, and leave the polling watchdog untouched
(caf_plan_watchdog == (0, 0)).  A hang here would be a protocol bug: every case runs once, under the suite's timeout."""

import os

import numpy as np
import pytest

from conftest import cn, qpsk

pytestmark = pytest.mark.gpu


def _case(seed=21, n=1024, m=600_000, f=96):
    rng = np.random.default_rng(seed)
    t = qpsk(rng, n)
    rx = cn(rng, m)
    d0, k0 = 345_678, -17
    rx[d0 : d0 + n] += (t * np.exp(2j * np.pi * k0 * np.arange(n) / n)).astype(np.complex64)
    return t, rx, np.arange(-f // 2, f // 2), (d0, k0)


@pytest.fixture(scope="module")
def fused_reference():
    from pydsproutines_amd import CAFPlan, asarray

    t, rx, bins, truth = _case()
    d_rx = asarray(rx)
    plan = CAFPlan(t, max_rx_len=rx.size, bins=bins, grid=t.size, engine="fused")
    res = plan.run(d_rx, surface=True)
    ref = {"surface": res.surface.get(), "row_max": res.row_max.get(), "row_arg": res.row_arg.get(),
           "peak": (int(res.peak_delay.get()[0]), int(bins[res.peak_freq.get()[0]]), float(res.peak_val.get()[0]))}
    assert ref["peak"][:2] == truth
    plan.close()
    return {"t": t, "rx": rx, "d_rx": d_rx, "bins": bins, "ref": ref}


@pytest.mark.timeout(300)
@pytest.mark.parametrize("wgs,tr_slots", [(8, 12), (64, 0), (64, 31), (255, 12), (512, 12), (512, 31), (256, 0)])
def test_residency_matrix_equals_fused_bit_for_bit(fused_reference, wgs, tr_slots):
    from pydsproutines_amd import CAFPlan

    c = fused_reference
    old = {k: os.environ.get(k) for k in ("CAF_PERSIST_WGS", "CAF_PERSIST_TR_SLOTS")}
    os.environ["CAF_PERSIST_WGS"], os.environ["CAF_PERSIST_TR_SLOTS"] = str(wgs), str(tr_slots)
    try:
        plan = CAFPlan(c["t"], max_rx_len=c["rx"].size, bins=c["bins"], grid=c["t"].size, engine="persistent")
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for surface in (True, False):   # tile role with the surface / the no-surface reduction
        res = plan.run(c["d_rx"], surface=surface)
        if surface:
            np.testing.assert_array_equal(res.surface.get(), c["ref"]["surface"])
            np.testing.assert_array_equal(res.row_arg.get(), c["ref"]["row_arg"])
        else:
            # (arguments may differ from the surface rule on float32 ties inside a hypothesis group: values may not)
            a, b = res.row_arg.get()[0], c["ref"]["row_arg"][0]
            rows = np.nonzero(a != b)[0]
            assert rows.size <= a.size // 20000
            s = c["ref"]["surface"][0]
            assert np.all(s[rows, a[rows]] == s[rows, b[rows]])
        np.testing.assert_array_equal(res.row_max.get(), c["ref"]["row_max"])
        pk = (int(res.peak_delay.get()[0]), int(c["bins"][res.peak_freq.get()[0]]), float(res.peak_val.get()[0]))
        assert pk == c["ref"]["peak"]
        assert plan.watchdog() == (0, 0)
    plan.close()


@pytest.mark.timeout(300)
def test_two_plans_on_two_streams(fused_reference):
    """Two persistent launches in flight at once (each sized for the whole chip, so their workgroups interleave and
    neither is fully resident): both finish, both are exact."""
    import ctypes as ct

    from pydsproutines_amd import CAFPlan, _lib, asarray

    c = fused_reference
    t2, rx2, bins2, truth2 = _case(seed=22, n=2048, m=500_000, f=64)
    d_rx2 = asarray(rx2)
    ref2 = CAFPlan(t2, max_rx_len=rx2.size, bins=bins2, grid=t2.size, engine="fused").run(d_rx2, surface=True)
    p1 = CAFPlan(c["t"], max_rx_len=c["rx"].size, bins=c["bins"], grid=c["t"].size, engine="persistent")
    p2 = CAFPlan(t2, max_rx_len=rx2.size, bins=bins2, grid=t2.size, engine="persistent")
    lib = _lib.load()
    s1, s2 = ct.c_void_p(), ct.c_void_p()
    _lib.check(lib.caf_stream_create(ct.byref(s1)))
    _lib.check(lib.caf_stream_create(ct.byref(s2)))
    _lib.check(lib.caf_stream_sync(None))  # inputs were uploaded on the default stream
    r1 = r2 = None
    for _ in range(3):
        r1 = p1.run(c["d_rx"], surface=True, stream=s1.value, out=r1)
        r2 = p2.run(d_rx2, surface=True, stream=s2.value, out=r2)
    _lib.check(lib.caf_stream_sync(s1))
    _lib.check(lib.caf_stream_sync(s2))
    np.testing.assert_array_equal(r1.surface.get(), c["ref"]["surface"])
    np.testing.assert_array_equal(r2.surface.get(), ref2.surface.get())
    assert (int(r2.peak_delay.get()[0]), int(bins2[r2.peak_freq.get()[0]])) == truth2
    assert p1.watchdog() == (0, 0) and p2.watchdog() == (0, 0)
    p1.close()
    p2.close()
    _lib.check(lib.caf_stream_destroy(s1))
    _lib.check(lib.caf_stream_destroy(s2))
