#!/bin/bash
# A/B of an environment switch of the persistent engine: parity tests + bench with VAR=0 and VAR=1
# usage: bash scripts/gpu_ab_env.sh CAF_PERSIST_TWREC
VAR=${1:?name of the environment switch}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/ab.log
for v in 0 1 0 1; do
  echo "== $VAR=$v" >> gpurun_out/ab.log
  env $VAR=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   ', j['engine'], round(j['value'],1), 'Msamples/s', round(j['ms_per_step'],2), 'ms; kernel', round(j['stages_ms_per_step']['spectral_conj_multiply'],3))" >> gpurun_out/ab.log
done
env $VAR=1 timeout -k 10 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -15 >> gpurun_out/ab.log
cat gpurun_out/ab.log
