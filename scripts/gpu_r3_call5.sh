#!/bin/bash
# round 3, call 5: step rounded to whole tiles + dead output quarters skipped: engine parity, then A/B/A/B vs round 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_persistent_protocol.py tests/test_gpu_fullsize.py tests/test_gpu_engine_fuzz.py -x -q -m gpu > gpurun_out/r3/t_engine2.log 2>&1
rc=$?
tail -5 gpurun_out/r3/t_engine2.log
[ $rc -ne 0 ] && exit $rc
: > gpurun_out/r3/ab_step.log
for lib in libcaf_r02 libcaf libcaf_r02 libcaf; do
  echo "== $lib" >> gpurun_out/r3/ab_step.log
  env CAF_LIBRARY=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   ', j['engine'], round(j['value'],1), 'Msamples/s', round(j['ms_per_step'],2), 'ms; kernel', round(j['stages_ms_per_step']['spectral_conj_multiply'],3), '; no surface', round(j['no_surface']['ms_per_step'],2))" >> gpurun_out/r3/ab_step.log || exit 1
done
cat gpurun_out/r3/ab_step.log
