#!/bin/bash
# full GPU suite, then the C3 / C4-share timings per engine
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/t16.log 2>&1
rc=$?
tail -25 gpurun_out/t16.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python scripts/time_configs.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/time_configs.log
