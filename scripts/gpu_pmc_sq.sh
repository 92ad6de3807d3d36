#!/bin/bash
# SQ counters of the engine's kernels (one rocprofv3 --pmc pass per counter group)
# usage: bash scripts/gpu_pmc_sq.sh [engine]   (auto = persistent)
set -e
ENGINE=${1:-auto}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_sq_$ENGINE
export PMC_SQ_OUT=$OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-side-figure --engine $ENGINE > $OUT/b$i.json 2> $OUT/e$i.err || { tail -5 $OUT/e$i.err; exit 1; }
done
cd $REPO
python3 - > $OUT/summary.txt <<'PY'
import csv, glob, collections, os
out = os.environ["PMC_SQ_OUT"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "caf::" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat $OUT/summary.txt
