"""Least-squares fit of the plan cost model of the mixed-radix per-delay kernel to scripts/sweep_mr_plans.py's timings:
   ms ~ threads(tpr) x ( sum over passes of  cnt_p x R_p x (w[R_p] + u [tpr not a multiple of 64])  +  c_pass )
with one weight per radix (first-pass radices separately: that pass loads from global memory and has no twiddles).
python scripts/fit_mr_model.py sweep.csv  -> the weights (normalised to w[16] = 40) and the ranking error per length."""
import sys
from collections import defaultdict

import numpy as np

rows = []
for line in open(sys.argv[1]):
    n, plan, tpr, ms, ok = line.strip().split(",")
    rows.append((int(n), [int(r) for r in plan.split("x")], int(tpr), float(ms), int(ok)))
assert all(r[4] for r in rows), "a plan found the wrong peak"
radices = sorted({r for _, rad, _, _, _ in rows for r in rad})
cols = {("f", r): i for i, r in enumerate(radices)}
cols.update({("l", r): len(radices) + i for i, r in enumerate(radices)})
npar = 2 * len(radices) + 2


def features(n, rad, tpr):
    rpw = max(1, 256 // tpr)
    threads = ((rpw * tpr + 63) // 64 * 64) / rpw
    f = np.zeros(npar)
    for i, r in enumerate(rad):
        cnt = -(-(n // r) // tpr)
        f[cols[("f" if i == 0 else "l", r)]] += cnt * r
        f[-1] += 1.0
        if tpr % 64:
            f[-2] += cnt * r
    return threads * f


A = np.array([features(n, rad, tpr) for n, rad, tpr, _, _ in rows])
y = np.array([ms for *_, ms, _ in rows])
# relative error: weight rows by 1 / y
w, *_ = np.linalg.lstsq(A / y[:, None], np.ones(len(y)), rcond=None)
scale = 40.0 / w[cols[("l", 16)]] if ("l", 16) in cols and w[cols[("l", 16)]] > 0 else 1.0
used = lambda k: bool(np.any(A[:, cols[k]]))  # noqa: E731
print("per point, first pass:", {r: round(float(w[cols[("f", r)]] * scale), 1) for r in radices if used(("f", r))})
print("per point, later pass:", {r: round(float(w[cols[("l", r)]] * scale), 1) for r in radices if used(("l", r))})
print("per pass: %.1f   per point of rows that do not fill whole waves: +%.1f" % (w[-1] * scale, w[-2] * scale))
pred = A @ w
print("relative error: rms %.3f, worst %.3f" % (np.sqrt(np.mean((pred / y - 1) ** 2)), np.max(np.abs(pred / y - 1))))
by_n = defaultdict(list)
for (n, rad, tpr, ms, _), p in zip(rows, pred):
    by_n[n].append((ms, p, rad, tpr))
for n, lst in sorted(by_n.items()):
    best = min(lst)
    pick = min(lst, key=lambda t: t[1])
    print("N=%6d  best %-16s/%-4d %7.3f ms | model picks %-16s/%-4d %7.3f ms (+%.1f %%)" % (
        n, "x".join(map(str, best[2])), best[3], best[0], "x".join(map(str, pick[2])), pick[3], pick[0], 100 * (pick[0] / best[0] - 1)))
