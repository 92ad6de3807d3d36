#!/usr/bin/env python3
"""Benchmark of the CAF hot path on MI355X (driver contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one batch of synthetic input that is already
resident in HBM: config C2 of BASELINE.json -- one 4096-sample template vs a 2^24-sample rx,
256 frequency-shift bins, full CAF surface (float32 [S][256]) + per-delay argmax + global peak.
With N > 1 ranks (one process per GPU, torch.distributed over RCCL) every rank evaluates its
own template against the replicated rx (weak scaling: template-sharded hypotheses) and the
per-template peak table (delay, freq, |peak|) is all-gathered inside the timed step.

--workload c4 runs BASELINE config 4 instead: 512 templates x 512 frequency bins against the same 2^24-sample
rx, templates block-sharded over the N ranks (STRONG scaling: the job is fixed, 512/N templates per GPU), peak
table only, the (delay, freq, |peak|) rows all-gathered over RCCL inside the timed step and checked on every
rank.  The default (and what the driver's fixed command line measures) stays C2, the configuration
BASELINE.json's metric is quoted on.

Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_TMPL = 4096
M_RX = 1 << 24
F_BINS = 256
D0 = 5_000_000
K0 = 37
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # ibid.: what streaming kernels reach


def make_inputs(torch, device, rank):
    """Synthetic complex64 IQ generated on the device (seeded): CN(0,1) noise + the rank's QPSK
    template planted at D0 with 0 dB SNR and a +K0-bin frequency offset."""
    g = torch.Generator(device=device)
    g.manual_seed(1)
    rx = torch.view_as_complex(torch.randn(M_RX, 2, generator=g, device=device, dtype=torch.float32) * (0.5**0.5))
    g.manual_seed(2 + rank)
    sym = torch.randint(0, 4, (N_TMPL,), generator=g, device=device)
    ph = (np.pi / 4) + (np.pi / 2) * sym.to(torch.float32)
    tmpl = torch.complex(torch.cos(ph), torch.sin(ph))
    n = torch.arange(N_TMPL, device=device, dtype=torch.float32)
    w = 2 * np.pi * K0 / N_TMPL
    tone = torch.complex(torch.cos(w * n), torch.sin(w * n))
    rx[D0 : D0 + N_TMPL] += tmpl * tone
    return rx.contiguous(), tmpl.contiguous()


def cpu_baseline(rx_host, tmpl_host, bins, budget_s=12.0):
    """The oracle (NumPy restatement of fastXcorr's per-delay CAF, xcorrRoutines.py:553-566) timed on
    one host core over a bounded sample of delays: chunks of 512 delays around the planted peak until
    the time budget is spent."""
    import oracle

    done = 0
    t0 = time.perf_counter()
    start = D0 - 4096
    while True:
        sh = np.arange(start + done, start + done + 512)
        oracle.caf_bins(tmpl_host, rx_host, bins, sh)
        done += 512
        el = time.perf_counter() - t0
        if el > budget_s or done >= (1 << 19):
            break
    return {
        "value": done / el / 1e6,
        "unit": "Msamples/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d consecutive delays starting at %d of the same rx/template, all 4096 FFT bins computed per "
        "delay then the 256 kept (reference algorithm), %.1f s" % (done, start, el),
        "us_per_delay": el / done * 1e6,
    }


def _libcaf_source_hash():
    import glob
    import hashlib

    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "pydsproutines_amd", "csrc", "*"))):
        if p.endswith((".hip", ".h", "Makefile")):
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(engine, surface_on, dom, args):
    """HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3
    runs) of scripts/gpu_profile_kernels.sh, as summarised in profiles/<round>/kernels_summary.json -- but only when that
    summary was taken from EXACTLY these kernel sources (hash of csrc/) and this configuration; otherwise null, never a
    stale constant.  bench.py cannot collect counters on itself: rocprofv3 has to own the process."""
    if args.log2_block or args.blocks_per_batch:
        return None, None
    wl = {("persistent", True): "c2_surface", ("persistent", False): "c2_nosurface", ("fused", True): "c2_fused",
          ("rocfft", True): "c2_rocfft"}.get((engine, surface_on))
    want = {"spectral_conj_multiply": {"persistent": "k_caf_persistent", "fused": "k_fused_caf", "rocfft": "k_spectral_mul"},
            "magsq_norm_argmax": {"fused": "k_transpose_norm_argmax", "rocfft": "k_magsq_norm_argmax"}}[dom].get(engine)
    if not wl or not want:
        return None, None
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "kernels_summary.json")), reverse=True):
        try:
            js = json.load(open(path))
            if js.get("libcaf_source_hash") != _libcaf_source_hash():
                continue
            for k in js["workloads"][wl]["kernels"]:
                if k["kernel"] == want and k["dispatches"] > 0:
                    per_launch = (k["fetch_bytes_per_call_x2"] + k["write_bytes_per_call"]) * k["calls"] / k["dispatches"]
                    if per_launch > 0:
                        return per_launch, os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def _host_cores():
    # threads actually usable: the scheduler affinity, capped at the 16-core share a one-GPU box gives
    # (os.cpu_count() reports all 256 host threads there); BENCH_CPU_WORKERS overrides
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return int(os.environ.get("BENCH_CPU_WORKERS", min(cores, 16)))


def cpu_baseline_threaded(rx_host, tmpl_host, budget_s=8.0):
    """SURVEY 8(d) baseline 2: the reference's threaded native correlator (IppXcorrFFT.cpp:94-178, threads strided
    over delays, full 4096-bin FFT per delay) as the plain-C restatement oracle/c/ippxcorrfft_port.c on all host
    cores, bounded sample of delays around the planted peak."""
    from oracle import cport

    cores = _host_cores()
    obj = cport.IppXcorrFFT(tmpl_host, cores)
    chunk = 4096 * cores
    done, pos = 0, 0
    start = D0 - 4096
    t0 = time.perf_counter()
    while True:
        if start + pos + chunk + tmpl_host.size > rx_host.size:
            pos = 0  # wrap: the host sample of rx is short
        obj.xcorr(rx_host, start + pos, start + pos + chunk, 1)
        done += chunk
        pos += chunk
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    return {
        "value": done / el / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d delays from %d on (wrapping), 4096-bin FFT + argmax per delay (scalar C FFT, not IPP), "
        "%d threads, %.1f s" % (done, start, cores, el),
        "us_per_delay_per_thread": el / done * 1e6 * cores,
    }


def cpu_baseline_same_algorithm(rx_host, tmpl_host, bins, budget_s=10.0):
    """BASELINE.md baseline B3: the algorithm the GPU runs (hypothesis-domain overlap-save CAF with shifted
    template spectra) as the oracle's scipy.fft restatement on ALL host cores, bounded sample of delays."""
    import oracle

    cores = _host_cores()
    blk = 1 << 16
    step = blk - tmpl_host.size + 1
    seg = 4 * step + tmpl_host.size - 1  # four overlap-save blocks per call
    done, pos = 0, 0
    t0 = time.perf_counter()
    while True:
        if pos + seg > rx_host.size:
            pos = 0
        oracle.caf_overlap_save(tmpl_host, rx_host[pos : pos + seg], bins, block=blk, workers=cores)
        done += 4 * step
        pos += 4 * step
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    return {
        "value": done / el / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d delays (x %d bins) in overlap-save blocks of 65536 via scipy.fft with workers=%d, %.1f s"
        % (done, len(bins), cores, el),
    }


C4_BINS = 512


def make_inputs_c4(torch, device, num_templates):
    """Config C4's synthetic input, identical on every rank (seeded device RNG): CN(0,1) noise of 2^24 samples with
    each of the QPSK templates planted once at 0 dB, at its own delay and on-grid frequency offset
    (sharding.c4_plant_plan).  Returns rx, templates (T, N), planted delays, planted bins."""
    from pydsproutines_amd import sharding

    g = torch.Generator(device=device)
    g.manual_seed(1)
    rx = torch.view_as_complex(torch.randn(M_RX, 2, generator=g, device=device, dtype=torch.float32) * (0.5**0.5))
    g.manual_seed(2)
    sym = torch.randint(0, 4, (num_templates, N_TMPL), generator=g, device=device)
    ph = (np.pi / 4) + (np.pi / 2) * sym.to(torch.float32)
    tm = torch.complex(torch.cos(ph), torch.sin(ph))
    delays, kbins = sharding.c4_plant_plan(num_templates, N_TMPL, C4_BINS, M_RX)
    n = torch.arange(N_TMPL, device=device, dtype=torch.float32)
    for i in range(num_templates):
        w = 2 * np.pi * int(kbins[i]) / N_TMPL
        rx[int(delays[i]) : int(delays[i]) + N_TMPL] += tm[i] * torch.complex(torch.cos(w * n), torch.sin(w * n))
    return rx.contiguous(), tm.contiguous(), delays, kbins


def run_c4(args, torch, dist, device, world, rank, rehearse):
    """BASELINE config 4: T templates x 512 bins, block-sharded over the ranks; one step = every rank's CAF over
    all delays for its templates (peak table only) + the all-gather of the table."""
    from pydsproutines_amd import CAFPlan, sharding
    from pydsproutines_amd.caf import CAFResult
    from pydsproutines_amd.devarray import DeviceArray

    T = args.templates
    rx, tm, delays, kbins = make_inputs_c4(torch, device, T)
    bins = np.arange(-C4_BINS // 2, C4_BINS // 2, dtype=np.int32)
    S = M_RX - N_TMPL + 1
    lo, hi = sharding.shard_range(T, world, rank)
    plan = CAFPlan(tm[lo:hi].cpu().numpy(), max_rx_len=M_RX, bins=bins, grid=N_TMPL, log2_block=args.log2_block,
                   blocks_per_batch=args.blocks_per_batch, engine=args.engine)
    # the engine's three per-template peak arrays are the rows of ONE torch tensor: the collective reads them in place
    cols = torch.zeros((3, hi - lo), dtype=torch.int32, device=device)
    res = CAFResult()
    res.peak_delay = DeviceArray((hi - lo,), np.int32, ptr=cols[0].data_ptr())
    res.peak_freq = DeviceArray((hi - lo,), np.int32, ptr=cols[1].data_ptr())
    res.peak_val = DeviceArray((hi - lo,), np.float32, ptr=cols[2].data_ptr())
    state = {}

    def compute_local(a, b):
        assert (a, b) == (lo, hi)
        plan.run(rx, surface=False, rows=False, peak=True, stream=torch.cuda.current_stream().cuda_stream, out=res)
        return cols.cpu() if rehearse else cols

    def step():
        state["table"] = sharding.sharded_peak_table(T, compute_local)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def check_table():
        tb = state["table"].cpu().numpy()
        ok = tb.shape == (T, 3) and np.array_equal(tb[:, 0], delays) and np.array_equal(bins[tb[:, 1]], kbins)
        if not ok:
            bad = [i for i in range(T) if tb[i, 0] != delays[i] or bins[tb[i, 1]] != kbins[i]][:5]
            raise SystemExit("rank %d: peak table wrong at templates %r" % (rank, bad))
        return tb

    for _ in range(max(args.warmup, 1)):
        step()
    fence()
    check_table()  # on every rank
    plan.profile(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    stages = plan.profile_get()
    plan.profile(False)
    tb = check_table()
    rank_ms, gather_ms = None, None
    if world > 1:
        elapsed, rank_ms = rank_times(torch, dist, elapsed, args.steps, world, rehearse, device)
        fence()
        tg = []
        for _ in range(5):  # the collective alone (the engine's three output arrays, read in place)
            t1 = time.perf_counter()
            sharding.all_gather_peak_columns(cols.cpu() if rehearse else cols, T)
            torch.cuda.synchronize()
            tg.append((time.perf_counter() - t1) * 1e3)
        gather_ms = {"first": tg[0], "median_of_5": float(np.median(tg))}
    if rank != 0:
        return
    dt = elapsed / args.steps
    B, step_len, nb = plan.block, plan.step, plan.blocks_per_batch
    ms, n = stages["spectral_conj_multiply"]
    nblk_total = -(-S // step_len)
    launches_per_step = max(1, n // max(args.steps, 1))
    t_local = hi - lo
    # work of this rank per step, spread over its launches
    flops_step = nblk_total * t_local * C4_BINS * (5.0 * B * np.log2(B) + 6.0 * B + 3.0 * step_len)
    avg_ms = ms / max(n, 1)
    tf = flops_step / launches_per_step / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    pv = tb[:, 2].copy().view(np.float32)
    out = {
        "metric": "CAF Msamples/s (C4: %d templates x %d freq bins, template-sharded, peak table)" % (T, C4_BINS),
        "value": T * S / dt / 1e6,
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "C4: %d templates x 4096 samples x %d on-grid freq bins vs one 2^24-sample rx (rx length is not "
                        "given in BASELINE.json: SURVEY 8d's choice), templates block-sharded over %d rank(s), per-template "
                        "(delay, freq, |peak|) table all-gathered inside the step and checked on every rank; value counts "
                        "template x delay pairs fully evaluated over all %d bins" % (T, C4_BINS, world, C4_BINS),
            "templates": T, "templates_per_gpu": t_local, "freq_bins": C4_BINS, "rx_len": M_RX, "delays": S,
            "block": B, "blocks_per_batch": nb, "parallelism": "template-shard x%d" % world,
        },
        "correlations_per_s": T * C4_BINS / dt,
        "caf_cells_per_s": T * C4_BINS * S / dt,
        "engine": plan.engine_used,
        "roofline": {
            "kernel": "k_caf_persistent, no-surface mode (multiply + LDS inverse FFT + |.|^2 + running per-delay maxima; "
                      "no |y|^2 tiles)" if plan.engine_used == "persistent" else plan.engine_used,
            "bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3, "traffic": None,
            "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step,
            "note": "f32 FFT butterflies on the vector ALUs (no MFMA instruction is used): 5*B*log2(B) + products + "
                    "|.|^2 per transform against the 157.3 TF f32 vector peak; HBM traffic of this mode is ~1/32 of the "
                    "surface mode's and is not the bound",
        },
        "peak_table_check": {"templates": T, "all_planted_peaks_exact_on_every_rank": True,
                             "min_peak_qf2": float(pv.min()), "max_peak_qf2": float(pv.max())},
        "stages_ms_per_step": {k: v[0] / args.steps for k, v in stages.items()},
    }
    if rank_ms is not None:
        out["per_rank_ms_per_step"] = {"min": min(rank_ms), "max": max(rank_ms), "ranks": rank_ms}
        out["peak_table_allgather_ms"] = gather_ms
    print(json.dumps(out))


def run_c2_freq_shard(args, torch, dist, device, world, rank, rehearse):
    """The metric's own configuration strong-scaled (second paragraph): ONE 4096-sample template x 256 bins, the
    bins block-distributed over the ranks (sharding.shard_bins), rx replicated; every rank evaluates all delays for its bins
    -- its column block (S, 256 / N) of the CAF surface stays in its own HBM -- and the only exchange is the all-gather of one
    (delay, bin, value) row per rank, reduced with the engine's tie rule (sharding.sharded_bin_peak)."""
    from pydsproutines_amd import CAFPlan, sharding
    from pydsproutines_amd.caf import CAFResult
    from pydsproutines_amd.devarray import DeviceArray

    rx, tmpl = make_inputs(torch, device, 0)  # the same template and rx on every rank
    bins = np.arange(-F_BINS // 2, F_BINS // 2, dtype=np.int32)
    lo, hi = sharding.shard_bins(F_BINS, world, rank)
    fl = hi - lo
    S = M_RX - N_TMPL + 1
    surface_on = not args.no_surface
    plan = CAFPlan(tmpl.cpu().numpy(), max_rx_len=M_RX, bins=bins[lo:hi], grid=N_TMPL, log2_block=args.log2_block,
                   blocks_per_batch=args.blocks_per_batch, engine=args.engine)
    res = CAFResult()
    t_surface = torch.empty((1, S, fl), dtype=torch.float32, device=device) if surface_on else None
    t_rowmax = torch.empty((1, S), dtype=torch.float32, device=device)
    t_rowarg = torch.empty((1, S), dtype=torch.int32, device=device)
    t_peak = torch.zeros(3, dtype=torch.int32, device=device)  # (delay, local bin index, value bits)
    if surface_on:
        res.surface = DeviceArray((1, S, fl), np.float32, ptr=t_surface.data_ptr())
    res.row_max = DeviceArray((1, S), np.float32, ptr=t_rowmax.data_ptr())
    res.row_arg = DeviceArray((1, S), np.int32, ptr=t_rowarg.data_ptr())
    res.peak_delay = DeviceArray((1,), np.int32, ptr=t_peak.data_ptr())
    res.peak_freq = DeviceArray((1,), np.int32, ptr=t_peak.data_ptr() + 4)
    res.peak_val = DeviceArray((1,), np.float32, ptr=t_peak.data_ptr() + 8)
    state = {}

    def compute_local(a, b):
        assert (a, b) == (lo, hi)
        plan.run(rx, surface=surface_on, rows=True, peak=True, stream=torch.cuda.current_stream().cuda_stream, out=res)
        return t_peak.cpu() if rehearse else t_peak

    def step():
        state["peak"], state["table"] = sharding.sharded_bin_peak(F_BINS, compute_local)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def check():
        d, f, v = state["peak"]
        if (int(d), int(bins[f])) != (D0, K0):
            raise SystemExit("rank %d: wrong peak (%d, bin %d), expected (%d, %d)" % (rank, d, int(bins[f]), D0, K0))
        return float(v)

    for _ in range(max(args.warmup, 1)):
        step()
    fence()
    check()  # on every rank
    plan.profile(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    stages = plan.profile_get()
    plan.profile(False)
    pv = check()
    rank_ms = None
    if world > 1:
        elapsed, rank_ms = rank_times(torch, dist, elapsed, args.steps, world, rehearse, device)
    if rank != 0:
        return
    dt = elapsed / args.steps
    B, step_len, nb = plan.block, plan.step, plan.blocks_per_batch
    ms, n = stages["spectral_conj_multiply"]
    avg_ms = ms / max(n, 1)
    nblk_total = -(-S // step_len)
    blocks_per_launch = nblk_total / max(1, -(-nblk_total // nb))
    tiles = -(-step_len // 64)
    # this rank's launch: |y|^2 tiles written and read back, its surface block and the per-delay results written
    alg = blocks_per_launch * (tiles * 64 * fl * 4.0 * (2.0 if surface_on else 0.0) + 8.0 * B * (fl / 64.0 + 1)) \
        + blocks_per_launch * step_len * (fl * (4.0 if surface_on else 0.0) + 12.0)
    out = {
        "metric": "CAF Msamples/s (4k template x 256 freq bins)",
        "value": S / dt / 1e6,
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("C2" if M_RX == 1 << 24 else "REHEARSAL SIZE (not C2)")
            + ": ONE 4096-sample template vs 2^%d-sample rx, its 256 on-grid freq bins block-distributed over %d rank(s) "
              "(%d per rank), " % (int(np.log2(M_RX)), world, fl)
            + ("every rank writes its column block f32[S][%d] of the CAF surface + its per-delay argmax; " % fl if surface_on
               else "per-delay argmax only; ")
            + "one (delay, bin, |peak|) row per rank all-gathered inside the step and reduced (lowest delay, then lowest bin "
              "on ties) on every rank; value counts delays fully evaluated over all 256 bins by the whole job",
            "templates": 1, "freq_bins": F_BINS, "freq_bins_per_gpu": fl, "rx_len": M_RX, "delays": S,
            "block": B, "blocks_per_batch": nb, "parallelism": "freq-shard x%d" % world,
        },
        "correlations_per_s": F_BINS / dt,
        "caf_cells_per_s": S * F_BINS / dt,
        "engine": plan.engine_used,
        "roofline": {"kernel": "k_caf_persistent (rank 0's launch: %d of the 256 bins)" % fl if plan.engine_used == "persistent"
                     else plan.engine_used, "bound": "hbm", "achieved": alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if avg_ms > 0 else 0.0, "traffic": None,
                     "avg_launch_ms": avg_ms, "alg_bytes_per_launch": alg},
        "peak_check": {"delay": D0, "bin": K0, "qf2": pv, "exact_on_every_rank": True},
        "stages_ms_per_step": {k: v[0] / args.steps for k, v in stages.items()},
    }
    if rank_ms is not None:
        out["per_rank_ms_per_step"] = {"min": min(rank_ms), "max": max(rank_ms), "ranks": rank_ms}
    print(json.dumps(out))


def rank_times(torch, dist, elapsed, steps, world, rehearse, device):
    """Every rank's own time for the timed steps: (max over ranks in s, per-rank ms per step) -- a straggler or a slow first
    collective shows as max >> min in the one line the driver keeps."""
    mine = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else device)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    vals = [float(t.item()) for t in allr]
    return max(vals), [v / steps * 1e3 for v in vals]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` started by hand (no torchrun): start N child processes of this script, one per
    LOCAL_RANK, with the rendezvous variables torch.distributed reads (127.0.0.1, a free port), wait for all of them,
    print rank 0's JSON line and return non-zero if any rank failed.  The parent never initialises the GPU (children are
    separate processes started with subprocess, never an exec of a process that has touched the device)."""
    import torch  # device_count() does not initialise the GPU on this image

    rehearse = os.environ.get("BENCH_REHEARSE_GLOO") == "1"
    have = torch.cuda.device_count()
    if not rehearse and have < n:
        print("bench.py --gpus %d: only %d GPU(s) visible (set BENCH_REHEARSE_GLOO=1 to rehearse the %d-rank code path "
              "on fewer GPUs over gloo)" % (n, have, n), file=sys.stderr)
        return 2
    # The rendezvous port is found by binding to port 0 and closing again, so another process can take it before rank 0
    # binds it.  ONLY that failure is retried (twice at most, on a fresh port): a child whose init_process_group raised exits
    # with RC_RENDEZVOUS before it has run anything on the GPU.  Every other failure -- a rank that found a wrong peak, died
    # from a GPU fault or a signal -- is reported at once with the rank codes and never run again.
    for attempt in range(3):
        rc = _launch_once(n, argv)
        if rc != RC_RENDEZVOUS or attempt == 2:
            return 1 if rc in (3, RC_RENDEZVOUS) else rc
        print("bench.py --gpus %d: the world did not form (attempt %d); trying another port" % (n, attempt + 1), file=sys.stderr)
    return 1


RC_RENDEZVOUS = 4  # exit code of a rank whose torch.distributed rendezvous failed (nothing has run on the GPU yet)


def _launch_once(n, argv):
    """One attempt of launch_ranks: 0 = rank 0's line printed, 1 = a rank failed, 3 = the world formed with the wrong size,
    RC_RENDEZVOUS = the world did not form (some rank's init_process_group failed and no rank failed any other way)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # All ranks are watched: a rank that dies before the rendezvous (bad device index, import error) would otherwise leave
    # the others waiting for the whole store timeout with the GPU held.  The first non-zero exit -- or the overall limit --
    # ends the rest.  Rank 0's stdout is drained by a thread so that a long JSON line cannot block it.
    import threading

    out0_parts = []
    reader = threading.Thread(target=lambda: out0_parts.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    limit = float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "3000"))
    t_start = time.monotonic()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.monotonic() - t_start > limit:
            failed = True
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=30)
    out0 = out0_parts[0] if out0_parts else ""
    codes = [p.wait() for p in procs]
    line = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if any(codes) or not line:
        print("bench.py --gpus %d: rank exit codes %r" % (n, codes), file=sys.stderr)
        # (the ranks ended by terminate() after the first failure report -15: they did not fail by themselves)
        own = [c for c in codes if c not in (0, -15)]
        return RC_RENDEZVOUS if own and all(c == RC_RENDEZVOUS for c in own) else 1
    if json.loads(line[-1]).get("n_gpus") != n:
        print("bench.py --gpus %d: the world that formed reports n_gpus=%r" % (n, json.loads(line[-1]).get("n_gpus")), file=sys.stderr)
        return 3
    print(line[-1])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2-block", type=int, default=0)
    ap.add_argument("--blocks-per-batch", type=int, default=0)
    ap.add_argument("--workload", default=os.environ.get("BENCH_WORKLOAD", "c2"), choices=["c2", "c4"],
                    help="c2: the metric's configuration (default); c4: 512 templates x 512 bins, template-sharded")
    ap.add_argument("--templates", type=int, default=512, help="c4 only: total number of templates (512 = config C4)")
    ap.add_argument("--shard", default="template", choices=["template", "freq"],
                    help="c2 with --gpus N: 'template' = every rank its own template (weak scaling, the default the driver "
                         "measures); 'freq' = ONE template, its 256 bins block-distributed over the ranks (strong scaling)")
    ap.add_argument("--engine", default="auto", choices=["auto", "persistent", "fused", "rocfft"])
    ap.add_argument("--no-surface", action="store_true", help="peak-only mode (no CAF surface written)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rx-log2", type=int, default=24,
                    help="log2 of the rx length (24 = BASELINE's configuration; smaller values are for rehearsals and "
                         "tests only and say so in config.workload)")
    ap.add_argument("--no-side-figure", action="store_true",
                    help="skip the untimed no-surface side run (profiling: only the measured configuration launches)")
    args = ap.parse_args()

    # --gpus N without a launcher: this process becomes the launcher.  It starts one child per rank (before anything
    # here touches the GPU), waits for them and forwards rank 0's JSON line; see launch_ranks().
    global M_RX, D0
    if args.rx_log2 != 24:
        if not 16 <= args.rx_log2 <= 26:
            raise SystemExit("--rx-log2 must be within 16..26")
        M_RX = 1 << args.rx_log2
        D0 = min(D0, M_RX // 3)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the world that formed has WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # Rehearsal switch (NOT the measured configuration): BENCH_REHEARSE_GLOO=1 runs the N-rank code path
    # with every rank on GPU 0 and the peak-table all-gather over gloo, so the multi-rank logic can be
    # exercised on a one-GPU box.  The driver's runs use one GPU per rank and RCCL.
    rehearse = os.environ.get("BENCH_REHEARSE_GLOO") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if rehearse:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=device)
        except tuple(getattr(dist, n) for n in ("DistNetworkError", "DistStoreError") if hasattr(dist, n)) as e:
            # the store could not bind / connect (port taken, peers never arrived): the one failure launch_ranks() starts
            # again for.  A backend (RCCL) error is not caught here and fails the run.
            print("rank %d: rendezvous failed: %r" % (rank, e), file=sys.stderr)
            sys.exit(RC_RENDEZVOUS)

    from pydsproutines_amd import CAFPlan, _lib
    from pydsproutines_amd.caf import CAFResult
    from pydsproutines_amd.devarray import DeviceArray

    _lib.check(_lib.load().caf_set_device(local_rank), "caf_set_device")
    if args.workload == "c4":
        run_c4(args, torch, dist, device, world, rank, rehearse)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.shard == "freq":
        run_c2_freq_shard(args, torch, dist, device, world, rank, rehearse)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    rx, tmpl = make_inputs(torch, device, rank)
    bins = np.arange(-F_BINS // 2, F_BINS // 2, dtype=np.int32)
    S = M_RX - N_TMPL + 1
    surface_on = not args.no_surface

    plan = CAFPlan(tmpl.cpu().numpy(), max_rx_len=M_RX, bins=bins, grid=N_TMPL, log2_block=args.log2_block,
                   blocks_per_batch=args.blocks_per_batch, engine=args.engine)
    # outputs live in torch-owned HBM so the peak table can go straight into the RCCL all-gather
    res = CAFResult()
    t_surface = torch.empty((1, S, F_BINS), dtype=torch.float32, device=device) if surface_on else None
    t_rowmax = torch.empty((1, S), dtype=torch.float32, device=device)
    t_rowarg = torch.empty((1, S), dtype=torch.int32, device=device)
    t_peak = torch.zeros(3, dtype=torch.int32, device=device)  # (delay, freq index, value bits)
    if surface_on:
        res.surface = DeviceArray((1, S, F_BINS), np.float32, ptr=t_surface.data_ptr())
    res.row_max = DeviceArray((1, S), np.float32, ptr=t_rowmax.data_ptr())
    res.row_arg = DeviceArray((1, S), np.int32, ptr=t_rowarg.data_ptr())
    res.peak_delay = DeviceArray((1,), np.int32, ptr=t_peak.data_ptr())
    res.peak_freq = DeviceArray((1,), np.int32, ptr=t_peak.data_ptr() + 4)
    res.peak_val = DeviceArray((1,), np.float32, ptr=t_peak.data_ptr() + 8)
    from pydsproutines_amd import sharding

    gathered = {}

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        plan.run(rx, surface=surface_on, rows=True, peak=True, stream=stream, out=res)
        if world > 1:
            # the only collective of the path: RCCL all-gather of the (delay, freq, |peak|) rows
            rows = t_peak.view(1, 3).cpu() if rehearse else t_peak.view(1, 3)
            gathered["table"] = sharding.all_gather_peak_table(rows, world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        step()
    fence()
    pk = t_peak.cpu().numpy()
    got = (int(pk[0]), int(bins[pk[1]]), float(pk[2:3].view(np.float32)[0]))
    if got[:2] != (D0, K0):
        raise SystemExit("rank %d: wrong peak %r, expected (%d, %d)" % (rank, got, D0, K0))

    plan.profile(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    stages = plan.profile_get()
    plan.profile(False)
    # the timed steps themselves must have produced the planted peak (not only the warm-up)
    pk = t_peak.cpu().numpy()
    got = (int(pk[0]), int(bins[pk[1]]), float(pk[2:3].view(np.float32)[0]))
    if got[:2] != (D0, K0):
        raise SystemExit("rank %d: wrong peak after the timed steps %r, expected (%d, %d)" % (rank, got, D0, K0))
    # ... and the surface they wrote must hold it: the peak's row (1 KB of the 17 GB) has its maximum at the planted bin, equal
    # to the reported peak value and to the per-delay result; a noise-only row far from it stays at the 1/N level
    if surface_on:
        row = t_surface[0, D0].cpu().numpy()
        far = t_surface[0, (D0 + S // 2) % S].cpu().numpy()
        if int(np.argmax(row)) != int(pk[1]) or row.max() != np.float32(got[2]) or row.max() != t_rowmax[0, D0].item() \
                or not (0.0 <= far.max() < 20.0 / N_TMPL):
            raise SystemExit("rank %d: the surface written by the timed steps does not hold the reported peak" % rank)

    # side figure (not the metric): the same job when only the per-delay argmax and the peak are wanted
    extra = None
    if world == 1 and surface_on and not args.no_side_figure:
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            plan.run(rx, surface=False, rows=True, peak=True, stream=stream, out=res)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            plan.run(rx, surface=False, rows=True, peak=True, stream=stream, out=res)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 5
        extra = {"ms_per_step": dt * 1e3, "value": S / dt / 1e6, "unit": "Msamples/s",
                 "what": "same workload without the CAF surface (per-delay argmax + peak only)"}

    # second side figure: the same surface written hypothesis-major ([F][S]: the FFT items write their own row segments, no
    # |y|^2 tiles, no tile role) -- what the host-returning entry points run since round 5 (xcorrRoutines._host_surface)
    extra_t = None
    if world == 1 and surface_on and not args.no_side_figure and plan.engine_used == "persistent" and plan.block == 16384:
        stream = torch.cuda.current_stream().cuda_stream
        t_surface = None  # (the delay-major surface is checked and done with: its 17 GB go back to torch's allocator)
        res.surface = None
        t_surface_t = torch.empty((1, F_BINS, S), dtype=torch.float32, device=device)
        res_t = CAFResult()
        res_t.surface_t = DeviceArray((1, F_BINS, S), np.float32, ptr=t_surface_t.data_ptr())
        res_t.row_max, res_t.row_arg = res.row_max, res.row_arg
        res_t.peak_delay, res_t.peak_freq, res_t.peak_val = res.peak_delay, res.peak_freq, res.peak_val
        for _ in range(2):
            plan.run(rx, surface_t=True, rows=True, peak=True, stream=stream, out=res_t)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            plan.run(rx, surface_t=True, rows=True, peak=True, stream=stream, out=res_t)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 5
        col = t_surface_t[0, :, D0].cpu().numpy()
        if int(np.argmax(col)) != int(pk[1]) or col.max() != np.float32(got[2]):
            raise SystemExit("rank %d: the hypothesis-major surface does not hold the reported peak" % rank)
        extra_t = {"ms_per_step": dt * 1e3, "value": S / dt / 1e6, "unit": "Msamples/s",
                   "what": "same workload with the surface written hypothesis-major f32[256][S] (caf_outputs2.d_surface_t) + "
                           "per-delay argmax + peak; the form behind the host-returning API, which transposes during the download"}
        del t_surface_t

    rank_ms, gather_ms = None, None
    if world > 1:
        # every rank's own time for the K steps (a straggler shows as max >> min) and what the collective alone costs
        elapsed, rank_ms = rank_times(torch, dist, elapsed, args.steps, world, rehearse, device)
        tb = gathered["table"].cpu().numpy()
        if not all(int(r[0]) == D0 and int(bins[r[1]]) == K0 for r in tb):
            raise SystemExit("gathered peak table is wrong: %r" % (tb,))
        rows = t_peak.view(1, 3).cpu() if rehearse else t_peak.view(1, 3)
        fence()
        tg = []
        for _ in range(5):
            t1 = time.perf_counter()
            sharding.all_gather_peak_table(rows, world)
            torch.cuda.synchronize()
            tg.append((time.perf_counter() - t1) * 1e3)
        gather_ms = {"first": tg[0], "median_of_5": float(np.median(tg))}

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        msamples = world * S / (elapsed / args.steps) / 1e6
        B, step_len, nb = plan.block, plan.step, plan.blocks_per_batch
        # algorithmic bytes per launch: one launch processes nb rx blocks
        persistent = plan.engine_used == "persistent"   # both fused stages as ONE work-queue launch
        fused = plan.engine_used in ("fused", "persistent")
        nblk_total = -(-S // step_len)
        blocks_per_launch = nblk_total / max(1, -(-nblk_total // nb))  # average (the last batch may be short)
        cells = blocks_per_launch * step_len * F_BINS
        if fused:
            # fused kernel: reads X once per hypothesis group, writes |y|^2 (4 B/cell, 64-delay tiles)
            tiles = -(-step_len // 64)
            mul_bytes = blocks_per_launch * (tiles * 64 * F_BINS * 4.0 + 8.0 * B * (F_BINS / 64.0 + 1))
            # transpose kernel: reads the tiles, writes the surface (+ row results)
            mag_bytes = cells * (4.0 + (4.0 if surface_on else 0.0)) + blocks_per_launch * step_len * (4.0 + 8.0)
            # standard FFT operation count + the X*H products + |.|^2
            flops_per_launch = blocks_per_launch * F_BINS * (5.0 * B * np.log2(B) + 6.0 * B + 3.0 * step_len)
        else:
            mul_bytes = blocks_per_launch * 8.0 * B * (F_BINS + 2)      # write F rows, read X and H0 once
            mag_bytes = cells * (8.0 + (4.0 if surface_on else 0.0)) + blocks_per_launch * step_len * (4.0 + 8.0)
        if persistent:
            mul_bytes, mag_bytes = mul_bytes + mag_bytes, 0.0   # one kernel does both stages
        st = {}
        for name, alg in (("spectral_conj_multiply", mul_bytes), ("magsq_norm_argmax", mag_bytes)):
            ms, n = stages[name]
            avg = ms / max(n, 1)
            st[name] = {"avg_ms": avg, "launches": n, "alg_bytes_per_launch": alg,
                        "achieved_GBs": alg / (avg * 1e-3) / 1e9 if avg > 0 else 0.0}
        if fused:
            st["spectral_conj_multiply"]["kernel"] = (
                "k_caf_persistent (multiply + LDS inverse FFT + |.|^2 on most CUs, transpose + QF^2 + argmax on the others)"
                if persistent else "k_fused_caf (multiply + LDS inverse FFT + |.|^2)")
            st["spectral_conj_multiply"]["alg_flops_per_launch"] = flops_per_launch
            st["spectral_conj_multiply"]["achieved_TFLOPs"] = (
                flops_per_launch / (st["spectral_conj_multiply"]["avg_ms"] * 1e-3) / 1e12
                if st["spectral_conj_multiply"]["avg_ms"] > 0 else 0.0)
            st["magsq_norm_argmax"]["kernel"] = "k_transpose_norm_argmax"
        for name in ("fft_forward", "fft_inverse(rocFFT)", "energy_prefix", "gather_blocks", "peak_reduce"):
            ms, n = stages[name]
            st[name] = {"avg_ms": ms / max(n, 1), "launches": n}
        stage_total = {k: v["avg_ms"] * v["launches"] / args.steps for k, v in st.items()}
        dom = max(("spectral_conj_multiply", "magsq_norm_argmax"), key=lambda k: stage_total[k])
        traffic, traffic_src = measured_traffic(plan.engine_used, surface_on, dom, args)
        combined_bytes = (st["spectral_conj_multiply"]["alg_bytes_per_launch"] + st["magsq_norm_argmax"]["alg_bytes_per_launch"])
        combined_ms = st["spectral_conj_multiply"]["avg_ms"] + st["magsq_norm_argmax"]["avg_ms"]
        out = {
            "metric": "CAF Msamples/s (4k template x 256 freq bins)",
            "value": msamples,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("C2" if M_RX == 1 << 24 else "REHEARSAL SIZE (not C2)")
                + ": 1x4096-sample template vs 2^%d-sample rx, 256 on-grid freq bins, " % int(np.log2(M_RX))
                + ("full CAF surface f32[S][256] + per-delay argmax + peak" if surface_on else "per-delay argmax + peak only"),
                "templates_per_gpu": 1, "freq_bins": F_BINS, "rx_len": M_RX, "delays": S,
                "block": B, "blocks_per_batch": nb, "parallelism": "template-shard x%d" % world,
            },
            "correlations_per_s": world * F_BINS / (elapsed / args.steps),
            "caf_cells_per_s": world * S * F_BINS / (elapsed / args.steps),
            "roofline": (
                {
                    # the fused FFT kernel is bound by the f32 vector ALUs / LDS, not by HBM: price it against
                    # the chip's f32 peak (157.3 TFLOP/s vector == f32 MFMA rate, MI355X_MICROARCH.md)
                    "kernel": "k_fused_caf", "bound": "mfma", "achieved": st[dom]["achieved_TFLOPs"],
                    "peak": 157.3, "unit": "TFLOP/s", "frac": st[dom]["achieved_TFLOPs"] / 157.3, "traffic": traffic,
                    "avg_launch_ms": st[dom]["avg_ms"], "alg_flops_per_launch": st[dom]["alg_flops_per_launch"],
                    "note": "f32 FFT butterflies on the vector ALUs (no MFMA instruction is used); the figure is "
                            "the standard 5*B*log2(B) FFT count + X*H products + |.|^2 against the 157.3 TF f32 peak",
                }
                if fused and not persistent and dom == "spectral_conj_multiply" else
                {
                    "kernel": st[dom].get("kernel", dom), "bound": "hbm", "achieved": st[dom]["achieved_GBs"],
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": st[dom]["achieved_GBs"] / HBM_PEAK_GBS,
                    # against what a copy kernel reaches on this chip (MI355X_MICROARCH.md: ~6.3 TB/s achievable of the 8 spec)
                    "frac_of_achievable": st[dom]["achieved_GBs"] / HBM_ACHIEVABLE_GBS, "achievable": HBM_ACHIEVABLE_GBS,
                    "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": st[dom]["avg_ms"],
                    "alg_bytes_per_launch": st[dom]["alg_bytes_per_launch"],
                    **({"also_TFLOPs": st[dom]["achieved_TFLOPs"], "also_frac_f32_peak": st[dom]["achieved_TFLOPs"] / 157.3,
                        "note": "one launch overlaps the ALU/LDS-bound FFT role (priced in also_TFLOPs against the "
                                "157.3 TF f32 vector peak) with the HBM-bound transpose role on different CUs"}
                       if persistent else {}),
                }
            ),
            "roofline_hbm_kernel": None if persistent else {
                "kernel": st["magsq_norm_argmax"].get("kernel", "k_magsq_norm_argmax"), "bound": "hbm",
                "achieved": st["magsq_norm_argmax"]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": st["magsq_norm_argmax"]["achieved_GBs"] / HBM_PEAK_GBS,
                "avg_launch_ms": st["magsq_norm_argmax"]["avg_ms"],
                "alg_bytes_per_launch": st["magsq_norm_argmax"]["alg_bytes_per_launch"],
            },
            "engine": plan.engine_used,
            "roofline_conjmul_plus_magsq": {
                "achieved": combined_bytes / (combined_ms * 1e-3) / 1e9 if combined_ms > 0 else 0.0,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (combined_bytes / (combined_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if combined_ms > 0 else 0.0,
            },
            "stages_ms_per_step": stage_total,
            "stages": st,
        }
        if extra is not None:
            out["no_surface"] = extra
        if extra_t is not None:
            out["surface_t"] = extra_t
        if rank_ms is not None:
            out["per_rank_ms_per_step"] = {"min": min(rank_ms), "max": max(rank_ms), "ranks": rank_ms}
            out["peak_table_allgather_ms"] = gather_ms
        if world == 1 and not args.no_cpu_baseline:
            rx_h, tm_h = rx[: min(M_RX, D0 + 560000)].cpu().numpy(), tmpl.cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(rx_h, tm_h, bins)
            out["cpu_baseline_threaded"] = cpu_baseline_threaded(rx_h, tm_h)
            out["cpu_baseline_same_algorithm"] = cpu_baseline_same_algorithm(rx_h, tm_h, bins)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
