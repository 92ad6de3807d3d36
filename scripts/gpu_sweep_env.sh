#!/bin/bash
# bench the persistent engine for several values of one environment switch
# usage: bash scripts/gpu_sweep_env.sh VAR v1 v2 ...
VAR=${1:?name}; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for v in "$@" "$@"; do
  echo "== $VAR=$v" >> gpurun_out/sweep.log
  env $VAR=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   ', j['engine'], round(j['value'],1), 'Msamples/s', round(j['ms_per_step'],2), 'ms; kernel', round(j['stages_ms_per_step']['spectral_conj_multiply'],3), '; no surface', round(j.get('no_surface',{}).get('ms_per_step',0),2))" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
