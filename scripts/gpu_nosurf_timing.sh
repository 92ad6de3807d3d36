#!/bin/bash
# no-surface mode (running maxima in the FFT role) against the tile path: C2 without the surface, C4 share
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/nosurf.log
for v in 0 1; do
  echo "== CAF_PERSIST_NOSURF=$v" >> gpurun_out/nosurf.log
  CAF_PERSIST_NOSURF=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-surface 2>&1 | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   C2 no surface:', j['engine'], round(j['value'],1), 'Msamples/s', round(j['ms_per_step'],2), 'ms; kernel', round(j['stages_ms_per_step']['spectral_conj_multiply'],3))" >> gpurun_out/nosurf.log
  CAF_PERSIST_NOSURF=$v timeout -k 10 300 python scripts/time_configs.py persistent 2>&1 | grep -v amdgpu.ids >> gpurun_out/nosurf.log
done
cat gpurun_out/nosurf.log
