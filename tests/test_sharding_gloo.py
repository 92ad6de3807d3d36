"""world_size-2 (and 3) gloo tests of the multi-GPU path: template block-sharding and the
all-gather of the per-template peak table, on CPU.  The same code runs over RCCL in bench.py."""

import os
import socket

import numpy as np
import pytest

from pydsproutines_amd import sharding


def test_shard_range_partitions_exactly():
    for n in (1, 7, 8, 64, 512, 513):
        for w in (1, 2, 3, 8):
            r = [sharding.shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1 and sizes == sharding.shard_counts(n, w)
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def test_pack_unpack_roundtrip():
    d = np.array([5_000_000, 3, 2**31 - 1], np.int32)
    f = np.array([37, -1, 0], np.int32)
    v = np.array([0.5, 1e-30, np.float32(np.pi)], np.float32)
    t = sharding.pack_peak_table(d, f, v)
    assert t.dtype == np.int32 and t.shape == (3, 3)
    d2, f2, v2 = sharding.unpack_peak_table(t)
    np.testing.assert_array_equal(d, d2)
    np.testing.assert_array_equal(f, f2)
    np.testing.assert_array_equal(v.view(np.int32), v2.view(np.int32))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_templates, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = sharding.shard_range(num_templates, world, rank)
        tids = np.arange(a, b)
        # what a rank's engine would report for its templates: (delay, freq index, value)
        local = sharding.pack_peak_table(1000 + 7 * tids, tids % 5, (tids + 1) / 1024.0)
        full = sharding.all_gather_peak_table(torch.from_numpy(local), num_templates)
        q.put((rank, full.numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,num_templates", [(2, 8), (2, 5), (3, 4)])
def test_all_gather_peak_table_gloo(world, num_templates):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_templates, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tids = np.arange(num_templates)
    want = sharding.pack_peak_table(1000 + 7 * tids, tids % 5, (tids + 1) / 1024.0)
    for r in range(world):
        np.testing.assert_array_equal(got[r], want)  # identical on every rank, template order preserved


# ---- config C4's step (sharding.sharded_peak_table) over gloo, with the oracle standing in for the GPU engine ----
C4_MINI = dict(num_templates=6, template_len=64, num_bins=8, rx_len=2400)


def _c4_mini_inputs():
    rng = np.random.default_rng(40)
    T, N, F, M = (C4_MINI[k] for k in ("num_templates", "template_len", "num_bins", "rx_len"))
    delays, kbins = sharding.c4_plant_plan(T, N, F, M, seed=41)
    tm = np.exp(1j * (np.pi / 4 + np.pi / 2 * rng.integers(0, 4, (T, N)))).astype(np.complex64)
    rx = ((rng.standard_normal(M) + 1j * rng.standard_normal(M)) / np.sqrt(2)).astype(np.complex64)
    for i in range(T):
        rx[delays[i] : delays[i] + N] += (tm[i] * np.exp(2j * np.pi * kbins[i] * np.arange(N) / N)).astype(np.complex64)
    return tm, rx, delays, kbins


def _oracle_compute_local(tm, rx, bins):
    """What a rank's engine reports for templates [lo, hi): the oracle's CAF, reduced to (delay, bin index, peak)."""
    import torch

    import oracle

    def compute(lo, hi):
        S = rx.size - tm.shape[1] + 1
        cols = np.zeros((3, hi - lo), np.int32)
        for j, t in enumerate(range(lo, hi)):
            surf = oracle.caf_bins(tm[t], rx, bins, np.arange(S))
            d, f = np.unravel_index(int(np.argmax(surf)), surf.shape)
            cols[:, j] = (d, f, np.float32(surf[d, f]).view(np.int32))
        return torch.from_numpy(cols)

    return compute


def _c4_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tm, rx, _, _ = _c4_mini_inputs()
        bins = np.arange(-C4_MINI["num_bins"] // 2, C4_MINI["num_bins"] // 2)
        table = sharding.sharded_peak_table(tm.shape[0], _oracle_compute_local(tm, rx, bins))
        q.put((rank, table.numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_c4_step_sharded_equals_single_process(world):
    """bench.py --workload c4 calls sharding.sharded_peak_table; here the same function runs over gloo with the
    oracle as the per-rank engine: the gathered table is identical on every rank, equals the one-process table
    and recovers every planted (delay, bin).  (world 4 with 6 templates: uneven shards 2/2/1/1.)"""
    import torch.multiprocessing as mp

    tm, rx, delays, kbins = _c4_mini_inputs()
    bins = np.arange(-C4_MINI["num_bins"] // 2, C4_MINI["num_bins"] // 2)
    single = sharding.sharded_peak_table(tm.shape[0], _oracle_compute_local(tm, rx, bins)).numpy()
    np.testing.assert_array_equal(single[:, 0], delays)
    np.testing.assert_array_equal(bins[single[:, 1]], kbins)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        np.testing.assert_array_equal(got[r], single)


# ---- strong scaling of ONE template: its frequency bins block-distributed (sharding.shard_bins / sharded_bin_peak) ----
BIN_MINI = dict(template_len=64, num_bins=16, rx_len=1500, d0=700, k0=3)


def _bin_mini_inputs():
    rng = np.random.default_rng(50)
    N, F, M = (BIN_MINI[k] for k in ("template_len", "num_bins", "rx_len"))
    t = np.exp(1j * (np.pi / 4 + np.pi / 2 * rng.integers(0, 4, N))).astype(np.complex64)
    rx = ((rng.standard_normal(M) + 1j * rng.standard_normal(M)) / np.sqrt(2)).astype(np.complex64)
    d0, k0 = BIN_MINI["d0"], BIN_MINI["k0"]
    rx[d0 : d0 + N] += (t * np.exp(2j * np.pi * k0 * np.arange(N) / N)).astype(np.complex64)
    rx[1200:1300] = 0  # zero-energy windows: NaN on every rank, index 0
    return t, rx, np.arange(-F // 2, F // 2)


def _oracle_bin_engine(t, rx, bins):
    """A rank's engine for bins [lo, hi): the oracle's surface restricted to them, as float32 like the GPU's, reduced the way
    the engine reduces it (first maximum over bins per delay, first maximum over delays)."""
    import torch

    import oracle

    S = rx.size - t.size + 1
    keep = {}

    def compute(lo, hi):
        with np.errstate(all="ignore"):
            surf = oracle.caf_bins(t, rx, bins[lo:hi], np.arange(S)).astype(np.float32)
        dead = np.all(np.isnan(surf), axis=1)
        rmax = np.where(dead, np.float32(np.nan), np.nanmax(np.where(np.isnan(surf), -1, surf), axis=1)).astype(np.float32)
        rarg = np.where(dead, 0, np.argmax(np.where(np.isnan(surf), -1, surf), axis=1)).astype(np.int32)
        d = int(np.nanargmax(rmax))
        keep["rows"] = (rmax, rarg)
        return torch.from_numpy(np.array([d, rarg[d], rmax[d : d + 1].view(np.int32)[0]], np.int32))

    return compute, keep


def _bin_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t, rx, bins = _bin_mini_inputs()
        compute, keep = _oracle_bin_engine(t, rx, bins)
        peak, table = sharding.sharded_bin_peak(bins.size, compute)
        q.put((rank, peak, table.copy(), keep["rows"][0].copy(), keep["rows"][1].copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_frequency_sharded_single_template_equals_single_process(world):
    """bench.py --shard freq calls sharding.sharded_bin_peak; here over gloo with the oracle as the per-rank engine: the
    reduced peak is identical on every rank and equals the one-process run bit for bit, and the per-delay (max, first argmax)
    merged from the ranks' parts (sharding.merge_bin_rows) equals the one-process per-delay results, NaN windows included."""
    import torch.multiprocessing as mp

    t, rx, bins = _bin_mini_inputs()
    compute, keep = _oracle_bin_engine(t, rx, bins)
    single_peak, _ = sharding.sharded_bin_peak(bins.size, compute)
    single_rows = keep["rows"]
    assert (single_peak[0], int(bins[single_peak[1]])) == (BIN_MINI["d0"], BIN_MINI["k0"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bin_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {g[0]: g[1:] for g in (q.get(timeout=180) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        peak, table = got[r][0], got[r][1]
        assert (peak[0], peak[1]) == (single_peak[0], single_peak[1])
        assert np.float32(peak[2]).view(np.int32) == np.float32(single_peak[2]).view(np.int32)
        np.testing.assert_array_equal(table, got[0][1])
    rmax, rarg = sharding.merge_bin_rows([got[r][2] for r in range(world)], [got[r][3] for r in range(world)], bins.size)
    np.testing.assert_array_equal(rmax, single_rows[0])  # (NaN == NaN here)
    np.testing.assert_array_equal(rarg, single_rows[1])
    assert np.isnan(rmax).any()


def test_reduce_bin_peaks_tie_rule():
    v = np.float32(0.5).view(np.int32)
    # equal values: the lowest delay wins; at equal delays the lowest GLOBAL bin (= the lower rank's)
    tb = np.array([[900, 1, v], [700, 0, v], [700, 2, v], [100, 0, np.float32(np.nan).view(np.int32)]], np.int32)
    d, f, val = sharding.reduce_bin_peaks(tb, 16)
    assert (d, f, float(val)) == (700, 4 + 0, 0.5)
    tb[0] = (700, 3, v)
    assert sharding.reduce_bin_peaks(tb, 16)[:2] == (700, 3)


# ---- bench.py --gpus N starts its own ranks (no torchrun needed) and never silently measures fewer GPUs ----
def _run_bench(extra_env, *argv, timeout=600):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus_n_without_enough_gpus_fails_loudly():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the launcher would really start two ranks")
    r = _run_bench({}, "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline")
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # a world that does not match --gpus is refused as well (a stale WORLD_SIZE=1 must not measure one GPU as "2")
    r = _run_bench({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2", "--steps", "1", "--warmup", "1")
    assert r.returncode != 0 and "world that formed" in (r.stderr + r.stdout)


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="CPU-only check of the launcher's failure path")
def test_bench_launcher_reports_failed_ranks_on_cpu():
    r = _run_bench({"BENCH_REHEARSE_GLOO": "1"}, "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline")
    assert r.returncode != 0 and "rank exit codes" in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_launches_its_own_ranks_rehearsal():
    """python bench.py --gpus 2 with no launcher around it: two ranks are started, share GPU 0 (rehearsal switch), gather
    the peak table over gloo, and rank 0's line says n_gpus = 2.  Small rx: this checks the launch path, not the rate."""
    import json

    r = _run_bench({"BENCH_REHEARSE_GLOO": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                   "--rx-log2", "20")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "template-shard x2" and "REHEARSAL" in j["config"]["workload"]
    assert j["value"] > 0 and j["steps"] == 2


@pytest.mark.gpu
def test_bench_gpus_2_frequency_shard_rehearsal():
    """python bench.py --gpus 2 --shard freq: ONE template, 128 bins per rank, the two (delay, bin, value) rows gathered over
    gloo (both ranks share GPU 0: rehearsal switch) and reduced to the planted peak on every rank; the line says strong."""
    import json

    r = _run_bench({"BENCH_REHEARSE_GLOO": "1"}, "--gpus", "2", "--shard", "freq", "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline", "--rx-log2", "20")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["parallelism"] == "freq-shard x2"
    assert j["config"]["freq_bins_per_gpu"] == 128 and j["peak_check"]["exact_on_every_rank"] and j["value"] > 0
