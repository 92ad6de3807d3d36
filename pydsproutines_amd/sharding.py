"""Multi-GPU sharding of the hypothesis set: one process per GPU,
``torch.distributed`` (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Template x frequency hypotheses are independent given rx, so templates are block-distributed
over the ranks and every rank evaluates all frequency bins of its templates against a replicated
rx.  There is no data-path collective; the only exchange is one all-gather of the per-template
peak table (int32 delay, int32 freq index, float32 |peak|^2 = 12 bytes per template).  The
reference has no multi-GPU code at all, so nothing here mirrors an upstream call pattern.

A SINGLE template can be split as well (the configuration BASELINE.json's metric is quoted on: one 4096-sample template x
256 bins on 1 / 2 / 4 / 8 GPUs = strong scaling): its frequency bins are block-distributed (``shard_bins``), every rank
evaluates all delays for its bins -- its column block of the CAF surface stays in its own HBM -- and the exchange is again
one all-gather, of one (delay, bin, value) row per rank, reduced with the engine's own tie rule (``reduce_bin_peaks``).
The nearest thing upstream is the thread-strided split over shifts of cython_ext/CyIppXcorrFFT/IppXcorrFFT.cpp:117.
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


def shard_bins(num_bins, world_size, rank):
    """Block distribution of one template's frequency bins: rank r evaluates bins [lo, hi) -- contiguous and in order, so a
    local hypothesis index f maps to the global index lo + f and "lowest index wins" carries across ranks."""
    return shard_range(num_bins, world_size, rank)


def reduce_bin_peaks(table, num_bins):
    """Combine the (world, 3) int32 table of per-rank peaks (delay, LOCAL bin index, float32 value bits) of a
    ``shard_bins`` split into the template's global peak (delay, GLOBAL bin index, value): the largest value; on equal values
    the lowest delay, then the lowest bin -- the order in which a one-process run meets them (the peak search keeps the first
    maximum over delays, the per-delay search the first over bins).  NaN values (a rank whose bins saw only zero-energy
    windows) never win.  Returns (delay int, bin int, value float32)."""
    table = np.ascontiguousarray(table, dtype=np.int32)
    world = table.shape[0]
    vals = table[:, 2].copy().view(np.float32)
    best = None
    for r in range(world):
        lo, _ = shard_bins(num_bins, world, r)
        v = vals[r]
        if np.isnan(v):
            continue
        key = (-float(v), int(table[r, 0]), lo + int(table[r, 1]))
        if best is None or key < best:
            best = key
    if best is None:
        return 0x7FFFFFFF, 0, np.float32(np.nan)
    return best[1], best[2], np.float32(-best[0])


def merge_bin_rows(row_max_parts, row_arg_parts, num_bins):
    """Per-delay (maximum, first argmax over bins) of the whole template from the per-rank parts of a ``shard_bins`` split
    (lists in rank order, each (S,)): the value of the first rank that attains the maximum, its local index shifted by the
    rank's first bin.  NaN (zero-energy window: NaN on every rank) stays NaN with index 0, as one process reports it."""
    world = len(row_max_parts)
    best = np.array(row_max_parts[0], dtype=np.float32, copy=True)
    arg = np.array(row_arg_parts[0], dtype=np.int32, copy=True)
    for r in range(1, world):
        lo, _ = shard_bins(num_bins, world, r)
        v = np.asarray(row_max_parts[r], dtype=np.float32)
        up = v > best  # strict: the lower bin keeps a tie; NaN compares false
        best = np.where(up, v, best)
        arg = np.where(up, lo + np.asarray(row_arg_parts[r], dtype=np.int32), arg)
    return best, arg


def sharded_bin_peak(num_bins, compute_local, group=None):
    """One step of the frequency-sharded CAF of ONE template (strong scaling of the metric's configuration).

    ``compute_local(lo, hi)`` evaluates bins [lo, hi) over all delays and returns the rank's peak as a (3,) int32 torch tensor
    (delay, local bin index, float32 value bits) on the device the group communicates from.  One all-gather of these rows;
    returns ((delay, global bin, value), the gathered (world, 3) table), identical on every rank."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_bins(num_bins, world, rank)
    row = compute_local(lo, hi)
    if tuple(row.shape) != (3,):
        raise ValueError("compute_local returned shape %r, expected (3,)" % (tuple(row.shape),))
    table = row.view(1, 3) if world == 1 else all_gather_peak_table(row.view(1, 3), world, group)
    tb = table.cpu().numpy()
    return reduce_bin_peaks(tb, num_bins), tb


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


def all_gather_peak_columns(cols, num_templates, group=None):
    """Same exchange as :func:`all_gather_peak_table` for the layout the engine writes: ``cols`` is this
    rank's (3, t_local) int32 tensor -- row 0 delays, row 1 frequency indices, row 2 the float32 peak values
    as bits, i.e. the three per-template output arrays of ``caf_plan_execute`` laid end to end in one
    allocation -- so the collective reads the engine's outputs in place.  Returns the (num_templates, 3) table."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    counts = shard_counts(num_templates, world)
    cmax = max(counts)
    if cols.shape[1] == cmax:
        send = cols.contiguous()
    else:
        send = torch.zeros((3, cmax), dtype=torch.int32, device=cols.device)
        send[:, : cols.shape[1]] = cols
    out = torch.empty((world * 3, cmax), dtype=torch.int32, device=cols.device)  # rank blocks stacked along dim 0
    dist.all_gather_into_tensor(out, send, group=group)
    keep = [out[3 * r : 3 * r + 3, : counts[r]].t() for r in range(world)]
    return torch.cat(keep, dim=0).contiguous()


def sharded_peak_table(num_templates, compute_local, group=None):
    """One step of the template-sharded peak search (BASELINE config C4).

    ``compute_local(lo, hi)`` evaluates templates [lo, hi) against the replicated rx and returns their peaks
    as a (3, hi - lo) int32 torch tensor (delay, frequency index, float32 value bits) on the device the
    process group communicates from (HBM for RCCL, host for gloo).  Templates are block-distributed over
    the ranks of ``group``; the only collective is the all-gather of the table, which is returned --
    identical on every rank -- as a (num_templates, 3) tensor.  Without an initialised process group the
    call is the single-GPU job.  bench.py --workload c4, the gloo tests and the one-GPU full-share test all
    go through this function."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_range(num_templates, world, rank)
    cols = compute_local(lo, hi)
    if tuple(cols.shape) != (3, hi - lo):
        raise ValueError("compute_local returned shape %r for templates [%d, %d)" % (tuple(cols.shape), lo, hi))
    if world == 1:
        return cols.t().contiguous()
    return all_gather_peak_columns(cols, num_templates, group)


def c4_plant_plan(num_templates, template_len, num_bins, rx_len, seed=4):
    """Where config C4's synthetic signals sit: template i is planted once, at delay d_i (evenly spread over the
    delay axis with a seeded jitter, never overlapping a neighbour) with an on-grid frequency offset of k_i
    bins, k_i in [-num_bins/2, num_bins/2).  Returns (delays int64[T], bins int32[T]); every rank derives the
    same plan from the seed."""
    rng = np.random.default_rng(seed)
    S = rx_len - template_len + 1
    pitch = S // num_templates
    if pitch < 2 * template_len:
        raise ValueError("rx too short to plant %d templates of %d samples apart" % (num_templates, template_len))
    jitter = rng.integers(0, pitch - template_len, num_templates)
    delays = pitch * np.arange(num_templates, dtype=np.int64) + jitter
    bins = rng.integers(-num_bins // 2, num_bins // 2, num_templates).astype(np.int32)
    return delays, bins
