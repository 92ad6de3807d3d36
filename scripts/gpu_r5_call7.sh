#!/bin/bash
set -o pipefail
O=gpurun_out/r05_call7
mkdir -p $O
export CAF_JIT_CACHE=off
timeout -k 10 900 python -m pytest tests/test_gpu_perdelay.py -x -q -k "jit" 2>&1 | tee $O/test_jit.log || exit 1
for pl in "5,16,15/96:1200" "12,10,10/128:1200" "16,15,5/80:1200" "5,20,14/112:1400" "5,20,10,5/256:5000" "20,25,10/256:5000" "12,8,16/128:1536" "8,15,16/128:1920" "6,25,20/192:3000" "10,20,18/256:3600" "16,20,25/512:8000" "10,10,6,20/704:12000" "12,8/8:96" "3,8,15/24:360"; do
  CAF_PDJ_PLAN="${pl%%:*}" timeout -k 10 120 python scripts/time_perdelay_mixed.py ${pl##*:} 2>&1 | grep "N=" | sed "s|^|plan ${pl%%:*}  |"
done | tee $O/timing_best_plans.log
for pl in "16,16,16/256:4096" "16,16,4/64:1024" "16,16/16:256" "8,16,8,16/1024:16384" "8,5,25/64:1000" "20,20,25/512:10000" "20,5/5:100"; do
  CAF_JIT_ALL=1 CAF_PDJ_PLAN="${pl%%:*}" timeout -k 10 120 python scripts/time_perdelay_mixed.py ${pl##*:} 2>&1 | grep "N=" | sed "s|^|plan ${pl%%:*}  |"
done | tee $O/timing_pow2_pow10_through_jit.log
