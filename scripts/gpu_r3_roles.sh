#!/bin/bash
# role split of the persistent launch (CAF_PERSIST_DEBUG=2: every 10th launch reports)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
LIB=${1:-libcaf}
env CAF_LIBRARY=$LIB CAF_PERSIST_DEBUG=2 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-side-figure > gpurun_out/r3/roles_$LIB.log 2>&1
grep -v "^{" gpurun_out/r3/roles_$LIB.log | tail -12
