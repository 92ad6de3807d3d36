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
