#!/bin/bash
# Per-kernel roofline evidence, run on the GPU box:
#   for every workload of scripts/profile_workloads.py: rocprofv3 kernel trace, then FETCH_SIZE and WRITE_SIZE in
#   PMC passes of their own (program directly after `--`; counters never combined with the other trace domains),
#   then scripts/summarize_kernels.py -> gpurun_out/prof_<tag>/kernels_summary.{txt,json}
# usage: bash scripts/gpu_profile_kernels.sh <tag> [workload ...]      (default: all workloads)
set -e
TAG=${1:-r03}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $REPO
WL="$@"
if [ -z "$WL" ]; then WL=$(python3 scripts/profile_workloads.py list); fi
export TMPDIR=/tmp
for w in $WL; do
  echo "== workload $w" | tee -a $OUT/progress.log
  mkdir -p $OUT/$w
  # (the trace pass repeats the workload so that the dominant kernel has >= 10 dispatches: median + minimum are reported)
  PROFILE_REPS_SCALE=${PROFILE_REPS_SCALE:-5} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w/trace -- python3 scripts/profile_workloads.py $w $OUT/$w/manifest.json > $OUT/$w/trace.log 2>&1 || { tail -20 $OUT/$w/trace.log; exit 1; }
  PROFILE_REPS_SCALE=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$w/pmc_fetch -- python3 scripts/profile_workloads.py $w $OUT/$w/manifest_pmc.json > $OUT/$w/fetch.log 2>&1 || { tail -20 $OUT/$w/fetch.log; exit 1; }
  PROFILE_REPS_SCALE=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/$w/pmc_write -- python3 scripts/profile_workloads.py $w > $OUT/$w/write.log 2>&1 || { tail -20 $OUT/$w/write.log; exit 1; }
  # where the waves' cycles go (SQ counters, a pass of their own; asked for with PROFILE_SQ=1: the small kernels)
  if [ -n "$PROFILE_SQ" ]; then
    PROFILE_REPS_SCALE=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/$w/pmc_sq -- python3 scripts/profile_workloads.py $w > $OUT/$w/sq.log 2>&1 || { tail -20 $OUT/$w/sq.log; exit 1; }
    # the LDS array's own cycles (a pass of their own): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE is the share of LDS-array cycles lost to
    # bank conflicts (MI355X_MICROARCH.md, LDS); SQ_ACTIVE_INST_LDS above counts wave cycles and is no denominator for it
    PROFILE_REPS_SCALE=1 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/$w/pmc_lds -- python3 scripts/profile_workloads.py $w > $OUT/$w/lds.log 2>&1 || { tail -20 $OUT/$w/lds.log; exit 1; }
  fi
done
python3 scripts/summarize_kernels.py $OUT > $OUT/kernels_summary.txt
cat $OUT/kernels_summary.txt
# the raw traces are large: keep the per-kernel statistics, drop dispatch-level files above 8 MB
find $OUT -name "*.csv" -size +8M -delete || true
find $OUT -name "*.db" -delete || true
