#!/bin/bash
# A/B/.../A/B/... of any number of library builds on one box: bash scripts/gpu_ab_libs_n.sh tag lib1 lib2 ... [-- bench args]
TAG=${1:?}; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" = "--" ] && shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
LOG=gpurun_out/r3/ab_$TAG.log
: > $LOG
for rep in 1 2; do
  for lib in "${LIBS[@]}"; do
    echo "== $lib" >> $LOG
    env CAF_LIBRARY=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>&1 | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   ', j['engine'], round(j['value'],1), 'Msamples/s', round(j['ms_per_step'],2), 'ms; kernel', round(j['stages_ms_per_step']['spectral_conj_multiply'],3), '; no surface', round(j.get('no_surface',{}).get('ms_per_step',0),2))" >> $LOG || exit 1
  done
done
cat $LOG
