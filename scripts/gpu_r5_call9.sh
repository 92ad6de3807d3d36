#!/bin/bash
set -o pipefail
O=gpurun_out/r05_call9
mkdir -p $O
export CAF_JIT_CACHE=off
timeout -k 10 900 python -m pytest tests/test_gpu_perdelay.py tests/test_gpu_api.py -x -q 2>&1 | tee $O/tests.log || exit 1
timeout -k 10 300 python scripts/time_perdelay_mixed.py 1200 1400 5000 1536 1920 3000 3600 8000 12000 96 360 1000 10000 1024 2048 2>&1 | tee $O/timing_perdelay_jit.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PROFILE_SQ=1 timeout -k 10 600 bash scripts/gpu_profile_kernels.sh r05jit perdelay_mixed_1200 perdelay_mixed_1400 perdelay_mixed_5000 > $O/profile.log 2>&1
tail -60 gpurun_out/prof_r05jit/kernels_summary.txt
