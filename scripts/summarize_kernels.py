#!/usr/bin/env python3
"""Merge the rocprofv3 outputs of scripts/gpu_profile_kernels.sh with the workloads' manifests:
per declared kernel -> dispatches, total / average duration, algorithmic bytes and flops per call, achieved GB/s against
the 8 TB/s HBM peak (and TFLOP/s against the 157.3 TF f32 vector peak where flops are declared), FETCH_SIZE (x2:
gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md) and WRITE_SIZE per call, and the ratio of
measured to algorithmic traffic.  Every other kernel of a run (rocFFT, copies) is listed with its time share.
Writes <dir>/kernels_summary.json and prints the text table."""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

HBM_PEAK, F32_PEAK = 8000.0, 157.3
SQ_COUNTERS = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT",
               "SQ_ACTIVE_INST_VALU", "SQ_WAVES")


def find(root, pattern):
    # (the NEWEST match: gpurun merges a call's files into gpurun_out/ beside those of earlier calls, whose process ids differ)
    r = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return max(r, key=os.path.getmtime) if r else None


def short(name):
    name = re.sub(r"\[clone.*$", "", name).strip()
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("caf::", "")
    name = name.split("(")[0]
    return name[-80:]


def load_trace(path):
    d = defaultdict(lambda: [0, 0.0, []])  # dispatches, total us, the individual durations
    if path:
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                us = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                d[k][0] += 1
                d[k][1] += us
                d[k][2].append(us)
    return d


def load_pmc(path, ctr):
    d = defaultdict(float)
    if path:
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == ctr:
                    d[short(row["Kernel_Name"])] += float(row["Counter_Value"])
    return d


def source_hash(repo):
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(repo, "pydsproutines_amd", "csrc", "*"))):
        if p.endswith((".hip", ".h", "Makefile")):
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def main(out):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    result = {"libcaf_source_hash": source_hash(repo), "hbm_peak_GBs": HBM_PEAK, "f32_peak_TFLOPs": F32_PEAK, "workloads": {}}
    for mpath in sorted(glob.glob(os.path.join(out, "*", "manifest.json"))):
        wdir = os.path.dirname(mpath)
        man = json.load(open(mpath))
        name = man["workload"]
        pmc_path = os.path.join(wdir, "manifest_pmc.json")  # the counter passes run the workload once (their own call counts)
        pmc_calls = {e["kernel"]: e["calls"] for e in json.load(open(pmc_path))["kernels"]} if os.path.exists(pmc_path) else {}
        trace = load_trace(find(os.path.join(wdir, "trace"), "*kernel_trace.csv"))
        fetch = load_pmc(find(os.path.join(wdir, "pmc_fetch"), "*counter_collection.csv"), "FETCH_SIZE")
        write = load_pmc(find(os.path.join(wdir, "pmc_write"), "*counter_collection.csv"), "WRITE_SIZE")
        sq_path = find(os.path.join(wdir, "pmc_sq"), "*counter_collection.csv")
        sq = {c: load_pmc(sq_path, c) for c in SQ_COUNTERS} if sq_path else None
        lds_path = find(os.path.join(wdir, "pmc_lds"), "*counter_collection.csv")
        lds = {c: load_pmc(lds_path, c) for c in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS")} if lds_path else None
        total_us = sum(v[1] for v in trace.values()) or 1.0
        print("=" * 150)
        print("workload %s   (all kernels of the run: %.1f ms)" % (name, total_us / 1e3))
        print("%-34s %6s %10s %10s | %9s %8s %7s | %8s %6s | %9s %9s %6s" % (
            "kernel (declared)", "disp", "total_ms", "avg_us", "alg_GB", "GB/s", "of8TB/s", "TFLOP/s", "of157", "FETCHx2GB", "WRITE_GB", "meas/alg"))
        used = set()
        wres = {"kernels": [], "others": []}
        for e in man["kernels"]:
            rx = re.compile(e["kernel"])
            ks = [k for k in trace if rx.search(k)]
            used.update(ks)
            disp = sum(trace[k][0] for k in ks)
            tus = sum(trace[k][1] for k in ks)
            calls = max(int(e["calls"]), 1)
            per_call_us = tus / calls
            ab, af = e["alg_bytes_per_call"], e["alg_flops_per_call"]
            gbs = ab / (per_call_us * 1e-6) / 1e9 if per_call_us > 0 else 0.0
            tfs = af / (per_call_us * 1e-6) / 1e12 if per_call_us > 0 else 0.0
            # KiB counters -> bytes per call; FETCH doubled
            pc = max(int(pmc_calls.get(e["kernel"], calls)), 1)
            fb = sum(fetch.get(k, 0.0) for k in ks) * 1024.0 * 2.0 / pc
            wb = sum(write.get(k, 0.0) for k in ks) * 1024.0 / pc
            durs = sorted(u for k in ks for u in trace[k][2])
            med_us = durs[len(durs) // 2] if durs else 0.0
            min_us = durs[0] if durs else 0.0
            ratio = (fb + wb) / ab if ab > 0 and (fb + wb) > 0 else None
            print("%-34s %6d %10.3f %10.1f | %9.3f %8.1f %7.3f | %8.2f %6.3f | %9.3f %9.3f %6s" % (
                e["kernel"][:34], disp, tus / 1e3, tus / max(disp, 1), ab / 1e9, gbs, gbs / HBM_PEAK, tfs, tfs / F32_PEAK,
                fb / 1e9, wb / 1e9, ("%.2f" % ratio) if ratio else "-"))
            print("      %s" % e["what"])
            print("      per dispatch: median %.1f us, minimum %.1f us over %d dispatches" % (med_us, min_us, disp))
            sqrec = None
            if sq:
                tot = {c: sum(sq[c].get(k, 0.0) for k in ks) for c in SQ_COUNTERS}
                wc = tot["SQ_WAVE_CYCLES"] or 1.0
                sqrec = {c: tot[c] for c in SQ_COUNTERS}
                # (SQ_ACTIVE_INST_LDS counts WAVE cycles with an LDS instruction in progress, SQ_LDS_BANK_CONFLICT cycles of the LDS
                #  array: their quotient -- printed up to round 4 as "of the LDS-active cycles", up to 425 % -- is no share of anything;
                #  the share of the array's cycles is conflict / SQ_LDS_IDX_ACTIVE, from the pmc_lds pass)
                ldsrec = None
                if lds:
                    lt = {c: sum(lds[c].get(k, 0.0) for k in ks) for c in lds}
                    ldsrec = dict(lt, conflict_share=lt["SQ_LDS_BANK_CONFLICT"] / max(lt["SQ_LDS_IDX_ACTIVE"], 1.0))
                print("      wave cycles: %.0f %% parked (s_waitcnt / barrier), %.0f %% issue-stalled, %.0f %% issuing (VALU %.0f %%, LDS %.0f %%)%s" % (
                          100 * tot["SQ_WAIT_ANY"] / wc, 100 * tot["SQ_WAIT_INST_ANY"] / wc, 100 * tot["SQ_ACTIVE_INST_ANY"] / wc,
                          100 * tot["SQ_ACTIVE_INST_VALU"] / wc, 100 * tot["SQ_ACTIVE_INST_LDS"] / wc,
                          ("; LDS array: %.0f %% of its active cycles are bank-conflict cycles (%.2e of %.2e), %.1f array cycles per LDS instruction"
                           % (100 * ldsrec["conflict_share"], ldsrec["SQ_LDS_BANK_CONFLICT"], ldsrec["SQ_LDS_IDX_ACTIVE"],
                              ldsrec["SQ_LDS_IDX_ACTIVE"] / max(ldsrec["SQ_INSTS_LDS"], 1.0))) if ldsrec else ""))
                sqrec["lds_array"] = ldsrec
            wres["kernels"].append({"kernel": e["kernel"], "what": e["what"], "matched": ks, "sq": sqrec, "dispatches": disp, "calls": calls,
                                    "total_ms": tus / 1e3, "avg_launch_us": tus / max(disp, 1), "median_launch_us": med_us,
                                    "min_launch_us": min_us, "per_call_us": per_call_us,
                                    "alg_bytes_per_call": ab, "achieved_GBs": gbs, "frac_hbm_peak": gbs / HBM_PEAK,
                                    "alg_flops_per_call": af, "achieved_TFLOPs": tfs, "frac_f32_peak": tfs / F32_PEAK,
                                    "fetch_bytes_per_call_x2": fb, "write_bytes_per_call": wb, "measured_over_algorithmic": ratio})
        rest = sorted(((k, v) for k, v in trace.items() if k not in used), key=lambda kv: -kv[1][1])
        if rest:
            print("   other kernels of the run (library transforms, copies, set-up):")
            for k, (n, t, _) in rest[:12]:
                print("      %-90s %6d disp %10.3f ms  %5.1f %%" % (k, n, t / 1e3, 100 * t / total_us))
                wres["others"].append({"kernel": k, "dispatches": n, "total_ms": t / 1e3, "share": t / total_us})
        result["workloads"][name] = wres
    json.dump(result, open(os.path.join(out, "kernels_summary.json"), "w"), indent=1)
    # registers / scratch / spills of every kernel and role, from the compiler's listing of these sources
    try:
        import subprocess

        txt = subprocess.run([sys.executable, os.path.join(repo, "scripts", "kernel_resources.py")], capture_output=True, text=True,
                             timeout=1200).stdout
        print("=" * 150)
        print("registers, scratch and spills (scripts/kernel_resources.py; scratch without spills in loops = the callee-saved "
              "registers of a noinline role, saved once per work item)")
        print(txt)
    except Exception as exc:  # (no compiler on the box: the table is produced where the library is built)
        print("kernel_resources.py not run: %r" % (exc,))


if __name__ == "__main__":
    main(sys.argv[1])
