"""Least-squares fit of the planner's cost model (caf_jit.hip, pdj_cost) to a plan sweep (scripts/sweep_pdj_plans.py), and the table of
measured-fastest plans (csrc/caf_pdj_tuned.inc):
    time per row  ~  threads a row occupies x ( sum over passes of  points per thread x w[position](radix) + b[position] per butterfly
                     + c[position] per pass )  +  a charge per row
w = A + B log2(radix) + C [radix not a power of two]; position = first / middle / last pass.  Plans with a first radix above 16 are
left out of the fit (they spill; the planner excludes them by rule).
Rows per workgroup: the default fills 256 threads; --rpw CSV (scripts/sweep_pdj_rpw.py) adds "xR" to a tuned plan where another count
measured more than 4 % faster (and the two sweeps agree on the default's time within 3 %).
usage: python scripts/fit_pdj_model.py profiles/r05/pdj_plan_sweep.csv [--rpw profiles/r05/pdj_rpw_sweep.csv]
                                       [--tuned pydsproutines_amd/csrc/caf_pdj_tuned.inc]"""
import collections
import csv
import math
import sys

import numpy as np

rows = []
for r in csv.reader(open(sys.argv[1])):
    try:
        n, rad, tpr, ms, ok = int(r[0]), tuple(int(x) for x in r[1].split("-")), int(r[2]), float(r[3]), int(r[4])
    except (ValueError, IndexError):
        continue
    if ok and ms == ms:
        rows.append((n, rad, tpr, ms))


def features(n, rad, tpr):
    rpw = max(1, 256 // tpr)
    threads = ((rpw * tpr + 63) // 64 * 64) / rpw
    f = np.zeros(16)
    for i, r in enumerate(rad):
        pos = 0 if i == 0 else (2 if i == len(rad) - 1 else 1)
        cnt = -(-(n // r) // tpr)
        pts = cnt * r
        f[pos * 3] += pts
        f[pos * 3 + 1] += pts * math.log2(r)
        f[pos * 3 + 2] += pts * (0 if (r & (r - 1)) == 0 else 1)
        f[9 + pos] += 1
        f[12 + pos] += cnt
    f = threads * f
    f[15] = 64.0
    return f


fit = [r for r in rows if r[1][0] <= 16]
A = np.array([features(n, rad, tpr) for n, rad, tpr, _ in fit])
y = np.array([ms * 1e3 for *_, ms in fit])  # ns per row
w = 1.0 / y  # relative errors
sol, *_ = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)
pred = A @ sol
print("plans %d (of %d timed), median |rel err| %.3f, 90th pct %.3f" % (len(fit), len(rows), np.median(np.abs(pred / y - 1)),
                                                                        np.quantile(np.abs(pred / y - 1), 0.9)))
s = sol * 1000
print("    static const double A[3] = {%.1f, %.1f, %.1f}, B[3] = {%.1f, %.1f, %.1f}, C[3] = {%.1f, %.1f, %.1f};" % (
    s[0], s[3], s[6], s[1], s[4], s[7], s[2], s[5], s[8]))
print("    static const double PASS[3] = {%.1f, %.1f, %.1f}, BFLY[3] = {%.1f, %.1f, %.1f};   // + 64.0 * %.1f per row" % (
    s[9], s[10], s[11], s[12], s[13], s[14], s[15]))
by = collections.defaultdict(list)
for (n, rad, tpr, ms) in rows:
    by[n].append((ms, float(features(n, rad, tpr) @ sol), rad, tpr))
rpw_best = {}
if "--rpw" in sys.argv:
    per = collections.defaultdict(list)
    for r in csv.reader(open(sys.argv[sys.argv.index("--rpw") + 1])):
        if len(r) >= 6 and r[4] != "nan" and int(r[5]):
            per[(int(r[0]), r[1], int(r[2]))].append((float(r[4]), int(r[3]), len(r) > 9 and r[9] == "default"))
    for k, v in per.items():
        dflt = [x for x in v if x[2]]
        if dflt and min(v)[0] < 0.96 * dflt[0][0]:
            rpw_best[k] = min(v) + (dflt[0][0],)
worst, tuned = 0.0, []
for n in sorted(by):
    best = min(by[n])
    cand = [c for c in by[n] if c[2][0] <= 16] or by[n]
    pick = min(cand, key=lambda t: t[1])
    loss = pick[0] / best[0] - 1
    worst = max(worst, loss)
    frac = 1e5 * 5 * n * math.log2(n) / (best[0] * 1e-3) / 157.3e12
    print("n=%5d fastest %-14s/%-4d %.3f ms per 1e5 rows (%.3f of the f32 peak) | model picks %-14s/%-4d %.3f ms (+%.1f %%)" % (
        n, "-".join(map(str, best[2])), best[3], best[0], frac, "-".join(map(str, pick[2])), pick[3], pick[0], 100 * loss))
    rb = rpw_best.get((n, "-".join(map(str, best[2])), best[3]))
    if rb and abs(rb[3] / best[0] - 1) < 0.03:  # (only where the two sweeps agree on the default's time: a noisy run proves nothing)
        frac = 1e5 * 5 * n * math.log2(n) / (rb[0] * 1e-3) / 157.3e12
        tuned.append('{%d, "%s/%dx%d"},  // %.3f ms per 1e5 delays, %.3f of the f32 peak (%d rows per workgroup: pdj_rpw_sweep.csv)' % (
            n, ",".join(map(str, best[2])), best[3], rb[1], rb[0], frac, rb[1]))
    else:
        tuned.append('{%d, "%s/%d"},  // %.3f ms per 1e5 delays, %.3f of the f32 peak' % (n, ",".join(map(str, best[2])), best[3], best[0], frac))
print("worst loss of the model's pick: %.1f %%" % (100 * worst))
if "--tuned" in sys.argv:
    path = sys.argv[sys.argv.index("--tuned") + 1]
    with open(path, "w") as f:
        f.write("// {length, \"radices/threads per row\"}: the fastest measured plan per length (scripts/sweep_pdj_plans.py on one MI355X, 32 plans per\n"
                "// length; %s).  Lengths not listed use the cost model (pdj_cost).  Generated by scripts/fit_pdj_model.py --tuned.\n" % sys.argv[1])
        f.write("\n".join(tuned) + "\n")
    print("wrote", path)
