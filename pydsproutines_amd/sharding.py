"""Multi-GPU sharding of the hypothesis set: one process per GPU,
``torch.distributed`` (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Template x frequency hypotheses are independent given rx, so templates are block-distributed
over the ranks and every rank evaluates all frequency bins of its templates against a replicated
rx.  There is no data-path collective; the only exchange is one all-gather of the per-template
peak table (int32 delay, int32 freq index, float32 |peak|^2 = 12 bytes per template).  The
reference has no multi-GPU code at all, so nothing here mirrors an upstream call pattern.
"""

import numpy as np


def shard_range(num_items, world_size, rank):
    """Block distribution: the first (num_items % world_size) ranks own one extra item.
    Returns (start, stop) of the rank's contiguous slice."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(int(num_items), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_counts(num_items, world_size):
    return [shard_range(num_items, world_size, r)[1] - shard_range(num_items, world_size, r)[0]
            for r in range(world_size)]


def pack_peak_table(delay, freq, val):
    """(T,) int32 delay, (T,) int32 freq index, (T,) float32 value -> (T, 3) int32 rows (value as bits)."""
    delay = np.asarray(delay, dtype=np.int32)
    freq = np.asarray(freq, dtype=np.int32)
    val = np.asarray(val, dtype=np.float32)
    return np.stack((delay, freq, val.view(np.int32)), axis=1)


def unpack_peak_table(table):
    table = np.ascontiguousarray(table, dtype=np.int32)
    return table[:, 0].copy(), table[:, 1].copy(), table[:, 2].copy().view(np.float32)


def all_gather_peak_table(local_rows, num_templates, group=None):
    """All-gather the per-template peak rows.  ``local_rows`` is this rank's (t_local, 3) int32 torch
    tensor (on the GPU for nccl/RCCL, on the CPU for gloo) in template order; returns the full
    (num_templates, 3) tensor, identical on every rank.  Uneven shards are padded to the largest
    shard so a single fixed-size collective suffices."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    counts = shard_counts(num_templates, world)
    cmax = max(counts)
    pad = torch.zeros((cmax, 3), dtype=torch.int32, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    out = torch.empty((world * cmax, 3), dtype=torch.int32, device=local_rows.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    keep = [out[r * cmax : r * cmax + counts[r]] for r in range(world)]
    return torch.cat(keep, dim=0)
