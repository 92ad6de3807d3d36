#!/bin/bash
# round 3, call 1: structural model of the FFT role + items of 64 / 32 / 16 hypotheses at C2 (A/B/A/B on one box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
timeout -k 10 300 scripts/ubench/fft_struct_model 256 3 > gpurun_out/r3/model.log 2>&1 || exit 1
cat gpurun_out/r3/model.log
: > gpurun_out/r3/hyp_sweep.log
for v in 64 32 16 64 32 16; do
  echo "== CAF_HYP_PER_WG=$v" >> gpurun_out/r3/hyp_sweep.log
  env CAF_HYP_PER_WG=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   ', j['engine'], round(j['value'],1), 'Msamples/s', round(j['ms_per_step'],2), 'ms; kernel', round(j['stages_ms_per_step']['spectral_conj_multiply'],3))" >> gpurun_out/r3/hyp_sweep.log || exit 1
done
cat gpurun_out/r3/hyp_sweep.log
