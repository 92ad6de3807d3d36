#!/bin/bash
# round 3: per-delay kernels -- parity tests, then A/B/A/B on one box
# (AB_LIBS: library builds via CAF_LIBRARY; AB_ENVS: "NAME=VALUE" settings, one run each, on libcaf.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
timeout -k 10 500 python -m pytest tests/test_gpu_perdelay.py tests/test_gpu_api.py tests/test_gpu_api_fuzz.py tests/test_gpu_reference_benchmarks.py -x -q > gpurun_out/r3/perdelay_tests.log 2>&1 || { tail -40 gpurun_out/r3/perdelay_tests.log; exit 1; }
tail -3 gpurun_out/r3/perdelay_tests.log
LOG=gpurun_out/r3/${AB_LOG:-ab_perdelay}.log
: > $LOG
for lib in ${AB_LIBS:-libcaf.so}; do
  for e in ${AB_ENVS:-X=0}; do
    echo "== $lib $e" >> $LOG
    env CAF_LIBRARY=$lib $e timeout -k 10 200 python scripts/time_perdelay.py 2>&1 | grep -v amdgpu.ids >> $LOG || exit 1
  done
done
cat $LOG
