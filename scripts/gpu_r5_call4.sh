#!/bin/bash
# Round 5: the run-time-compiled per-delay kernel: parity, timing against the prebuilt kernel; host-returning surface calls old / new
set -o pipefail
O=gpurun_out/r05_call4
mkdir -p $O
export CAF_JIT_CACHE=off
timeout -k 10 900 python -m pytest tests/test_gpu_perdelay.py -x -q -k "jit" 2>&1 | tee $O/test_jit.log || exit 1
timeout -k 10 600 python scripts/time_perdelay_mixed.py 2>&1 | tee $O/timing_perdelay_mixed.log || exit 1
timeout -k 10 900 python scripts/time_host_surface.py new 2>&1 | tee $O/timing_host_surface.log || exit 1
timeout -k 10 900 python scripts/time_host_surface.py old 2>&1 | tee -a $O/timing_host_surface.log
