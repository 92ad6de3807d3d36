#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 120 python scripts/dbg_persist2.py > gpurun_out/d10.log 2>&1
echo "rc=$?" >> gpurun_out/d10.log
cat gpurun_out/d10.log
