#!/bin/bash
# round 3: k_block_spectra32 variants (library builds via CAF_LIBRARY) under rocprofv3's kernel trace; tests first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_engine.py -x -q -k "long_template or around_the_fused" > gpurun_out/r3/bs32_tests.log 2>&1 || { tail -30 gpurun_out/r3/bs32_tests.log; exit 1; }
tail -2 gpurun_out/r3/bs32_tests.log
for lib in ${AB_LIBS:-libcaf.so libcaf_a.so}; do
  export CAF_LIBRARY=$lib
  timeout -k 10 200 python -m pytest tests/test_gpu_engine.py -x -q -k "long_template or around_the_fused" > gpurun_out/r3/bs32_tests_$lib.log 2>&1 || { tail -30 gpurun_out/r3/bs32_tests_$lib.log; exit 1; }
  rm -rf gpurun_out/r3/bs32_$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/bs32_$lib -- python3 scripts/profile_workloads.py c2_long_template > gpurun_out/r3/bs32_$lib.log 2>&1 || { tail gpurun_out/r3/bs32_$lib.log; exit 1; }
  echo "== $lib"
  grep -h "k_block_spectra32\|k_caf_persistent" $(find gpurun_out/r3/bs32_$lib -name "*kernel_stats.csv")
  find gpurun_out/r3/bs32_$lib -name "*kernel_trace.csv" -delete
done
