#!/bin/bash
# kernel durations of one profile workload under rocprofv3's kernel trace, per library build (CAF_LIBRARY)
# usage: bash scripts/gpu_ab_kernel_trace.sh <workload> <kernel-regex> lib1 lib2 ...
W=${1:?}; K=${2:?}; shift 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
for rep in 1 2; do
for lib in "$@"; do
  export CAF_LIBRARY=$lib
  rm -rf gpurun_out/r3/kt_$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/kt_$lib -- python3 scripts/profile_workloads.py $W > gpurun_out/r3/kt_$lib.log 2>&1 || { tail gpurun_out/r3/kt_$lib.log; exit 1; }
  echo "== $lib"
  grep -h -E "$K" $(find gpurun_out/r3/kt_$lib -name "*kernel_stats.csv") | cut -c1-60,100-
  find gpurun_out/r3/kt_$lib -name "*kernel_trace.csv" -delete
done
done
