#!/bin/bash
set -o pipefail
O=gpurun_out/r05_call10
mkdir -p $O
export CAF_JIT_CACHE=off
CAF_JIT_ALL=1 timeout -k 10 600 python scripts/time_perdelay_mixed.py 64 128 256 512 1024 2048 4096 8192 16384 100 1000 10000 2>&1 | tee $O/timing_pow2_pow10_through_jit.log
