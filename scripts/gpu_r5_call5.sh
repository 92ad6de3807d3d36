#!/bin/bash
set -o pipefail
O=gpurun_out/r05_call5
mkdir -p $O
export CAF_JIT_CACHE=off
timeout -k 10 900 python -m pytest tests/test_gpu_perdelay.py -x -q -k "jit" 2>&1 | tee $O/test_jit.log || exit 1
timeout -k 10 600 python scripts/time_perdelay_mixed.py 2>&1 | tee $O/timing_perdelay_mixed.log || exit 1
for plan in "16,15,5/80" "16,15,5/96" "16,15,5/128" "15,16,5/80" "5,16,15/80" "20,12,5/64" "20,10,6/64" "16,5,15/80" "25,8,6/64" "15,10,8/80" "20,20,3/64"; do
  echo "== 1200 plan $plan"; CAF_PDJ_PLAN=$plan timeout -k 10 120 python scripts/time_perdelay_mixed.py 1200 2>&1 | grep "N="
done | tee $O/sweep_1200.log
for plan in "25,20,10/250" "25,20,10/256" "20,25,10/256" "10,20,25/256" "20,10,5,5/256" "10,10,10,5/256" "25,10,20/256" "16,5,5,5,... " ; do
  case "$plan" in *...*) continue;; esac
  echo "== 5000 plan $plan"; CAF_PDJ_PLAN=$plan timeout -k 10 120 python scripts/time_perdelay_mixed.py 5000 2>&1 | grep "N="
done | tee $O/sweep_5000.log
