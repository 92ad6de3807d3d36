#!/bin/bash
# round 3, call 2: can the tile role share a CU with the FFT role (co-residency model)?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
timeout -k 10 300 scripts/ubench/coresident_model 1024 131072 > gpurun_out/r3/coresident.log 2>&1
echo "exit $?" >> gpurun_out/r3/coresident.log
cat gpurun_out/r3/coresident.log
