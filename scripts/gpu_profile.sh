#!/bin/bash
# Run on the GPU box: kernel-trace stats + PMC (FETCH_SIZE / WRITE_SIZE in separate passes) of bench.py.
# usage: bash scripts/gpu_profile.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-figure "$@" > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-side-figure "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err || { tail -20 $OUT/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-side-figure "$@" > $OUT/bench_write.json 2> $OUT/write.err || { tail -20 $OUT/write.err; exit 1; }
cd $REPO
python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
# keep only the small summaries (the raw traces are large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
