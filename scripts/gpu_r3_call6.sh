#!/bin/bash
# round 3, call 6: complex-QF rows from the persistent engine: parity + C3 literal timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_api.py tests/test_gpu_api_fuzz.py -x -q -m gpu > gpurun_out/r3/t_cqf.log 2>&1
rc=$?
tail -15 gpurun_out/r3/t_cqf.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/time_tcc_literal.py > gpurun_out/r3/tcc_literal.log 2>&1
cat gpurun_out/r3/tcc_literal.log
