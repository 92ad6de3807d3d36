"""Least-squares fit of the planner's cost model (caf_jit.hip, pdj_cost) to a plan sweep (scripts/sweep_pdj_plans.py):
    time per row  ~  threads a row occupies x ( sum over passes of  points per thread x w[position][radix]  +  c[position] )
position = first / middle / last pass.  Prints the weights as C++ tables and how far the model's pick is from the fastest plan.
usage: python scripts/fit_pdj_model.py profiles/r05/pdj_plan_sweep.csv"""
import collections
import csv
import sys

import numpy as np

RADICES = (2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 15, 16, 18, 20, 25)
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
    f = np.zeros(3 * len(RADICES) + 3)
    for i, r in enumerate(rad):
        pos = 0 if i == 0 else (2 if i == len(rad) - 1 else 1)
        pts = -(-(n // r) // tpr) * r
        f[pos * len(RADICES) + RADICES.index(r)] += pts
        f[3 * len(RADICES) + pos] += 1
    return threads * f


A = np.array([features(n, rad, tpr) for n, rad, tpr, _ in rows])
y = np.array([ms * 1e3 for *_, ms in rows])  # ns per row (x 1e5 rows = the measured ms)
# relative errors matter: weight each row by 1 / y
w = 1.0 / y
sol, *_ = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)
pred = A @ sol
print("plans %d, median |rel err| %.3f, 90th pct %.3f" % (len(rows), np.median(np.abs(pred / y - 1)), np.quantile(np.abs(pred / y - 1), 0.9)))
names = ("first", "middle", "last")
for p in range(3):
    print("// %s pass: weight per point by radix" % names[p])
    print("{" + ", ".join("{%d, %.1f}" % (r, sol[p * len(RADICES) + i]) for i, r in enumerate(RADICES)) + "}")
print("// per-pass charges (first, middle, last):", ", ".join("%.1f" % v for v in sol[3 * len(RADICES):]))
by = collections.defaultdict(list)
for (n, rad, tpr, ms), p in zip(rows, pred):
    by[n].append((ms, p, rad, tpr))
worst = 0.0
for n in sorted(by):
    best = min(by[n])
    pick = min(by[n], key=lambda t: t[1])
    loss = pick[0] / best[0] - 1
    worst = max(worst, loss)
    print("n=%5d fastest %s/%d %.3f ms | model picks %s/%d %.3f ms (+%.1f %%)" % (n, "-".join(map(str, best[2])), best[3], best[0],
                                                                              "-".join(map(str, pick[2])), pick[3], pick[0], 100 * loss))
print("worst loss of the model's pick: %.1f %%" % (100 * worst))
