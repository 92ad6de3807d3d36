#!/bin/bash
# gpurun_out/prof_<tag> (scratch, merged back from the GPU box) -> profiles/<tag> (tracked): the kernel summary and, per workload,
# the manifests and rocprofv3's own kernel statistics.   usage: bash scripts/collect_profiles.sh r04
set -e
TAG=${1:-r04}
SRC=gpurun_out/prof_$TAG
DST=profiles/$TAG
mkdir -p $DST/rocprofv3
python3 scripts/summarize_kernels.py $SRC > $SRC/kernels_summary.txt
cp $SRC/kernels_summary.txt $SRC/kernels_summary.json $DST/
for w in $SRC/*/; do
  n=$(basename $w)
  [ -f $w/manifest.json ] || continue
  mkdir -p $DST/rocprofv3/$n
  cp $w/manifest.json $DST/rocprofv3/$n/
  [ -f $w/manifest_pmc.json ] && cp $w/manifest_pmc.json $DST/rocprofv3/$n/
  f=$(ls -t $(find $w/trace -name "*kernel_stats.csv") 2>/dev/null | head -1)  # the newest (earlier calls' files stay beside it)
  [ -n "$f" ] && cp $f $DST/rocprofv3/$n/trace_kernel_stats.csv
done
echo "collected into $DST"
