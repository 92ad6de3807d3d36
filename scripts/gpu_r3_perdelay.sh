#!/bin/bash
# round 3: per-delay kernel epilogue -- parity tests, then A/B/A/B of two library builds on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
timeout -k 10 500 python -m pytest tests/test_gpu_perdelay.py tests/test_gpu_api.py tests/test_gpu_api_fuzz.py tests/test_gpu_reference_benchmarks.py -x -q > gpurun_out/r3/perdelay_tests.log 2>&1 || { tail -40 gpurun_out/r3/perdelay_tests.log; exit 1; }
tail -3 gpurun_out/r3/perdelay_tests.log
LOG=gpurun_out/r3/ab_perdelay_epilogue.log
: > $LOG
for lib in libcaf_prev.so libcaf.so libcaf_prev.so libcaf.so; do
  echo "== $lib" >> $LOG
  CAF_LIBRARY=$lib timeout -k 10 200 python scripts/time_perdelay.py 2>&1 | grep -v amdgpu.ids >> $LOG || exit 1
done
cat $LOG
