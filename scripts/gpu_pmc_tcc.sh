#!/bin/bash
# L2 <-> fabric counters of the persistent launch (C2, full surface; with 64- and 16-hypothesis FFT items; and the
# no-surface launch for contrast): requests, DRAM requests, request LEVEL (occupancy integral -> average latency =
# LEVEL / requests), write stalls.  One rocprofv3 --pmc pass per counter group, kernel trace only.
# usage: bash scripts/gpu_pmc_tcc.sh
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_tcc
export PMC_TCC_OUT=$OUT
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in h64 h16; do
  if [ $cfg = h16 ]; then export CAF_HYP_PER_WG=16; else unset CAF_HYP_PER_WG; fi
  i=0
  for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_LEVEL_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUBBLE_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$cfg/g$i -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/$cfg.b$i.json 2> $OUT/$cfg.e$i.err || { tail -5 $OUT/$cfg.e$i.err; exit 1; }
  done
done
cd $REPO
python3 - > $OUT/summary.txt <<'PY'
import csv, glob, collections, os
out = os.environ["PMC_TCC_OUT"]
for cfg in ("h64", "h16"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/" + cfg + "/g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            if "k_caf_persistent" in k:
                acc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== FFT items of %s hypotheses" % cfg[1:])
    for (k, g), d in acc.items():
        n = max(len(v) for v in d.values())
        print(" %s  (%d dispatches per pass: [0] = with the surface, [1] = without, as bench.py runs them)" % (k, n))
        for c, v in sorted(d.items()):
            print("   %-40s %s" % (c, "  ".join("%16.0f" % x for x in v)))
        def avg(a, b):
            return "  ".join("%10.1f" % (x / y if y else 0.0) for x, y in zip(d.get(a, []), d.get(b, [])))
        print("   read latency  = RDREQ_LEVEL / RDREQ  [cycles]:", avg("TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum"))
        print("   write latency = WRREQ_LEVEL / WRREQ  [cycles]:", avg("TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_sum"))
PY
cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +2M -delete || true
