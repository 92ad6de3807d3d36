#!/bin/bash
# round 3: in-LDS block spectra for the 32768-point role -- the long-template tests, then timing against the
# gather + rocFFT + parity-major form it replaces (CAF_FWD_ROCFFT=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
timeout -k 10 400 python -m pytest tests/test_gpu_engine.py -x -q -k "long_template or around_the_fused or complex_qf or surface" > gpurun_out/r3/fwd32_tests.log 2>&1 || { tail -30 gpurun_out/r3/fwd32_tests.log; exit 1; }
tail -3 gpurun_out/r3/fwd32_tests.log
timeout -k 10 200 python scripts/time_long_template.py > gpurun_out/r3/long_template_ldsfwd.log 2>&1 &&
CAF_FWD_ROCFFT=1 timeout -k 10 200 python scripts/time_long_template.py > gpurun_out/r3/long_template_rocfftfwd.log 2>&1 &&
timeout -k 10 200 python scripts/time_long_template.py > gpurun_out/r3/long_template_ldsfwd_2.log 2>&1
tail -n 12 gpurun_out/r3/long_template_ldsfwd.log gpurun_out/r3/long_template_rocfftfwd.log gpurun_out/r3/long_template_ldsfwd_2.log
