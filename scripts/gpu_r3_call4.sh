#!/bin/bash
# round 3, call 4: tile-first slots after the FFT role got faster
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
bash scripts/gpu_sweep_env.sh CAF_PERSIST_TR_SLOTS 12 13 14 15 16 > gpurun_out/r3/tr_slots.log 2>&1
cat gpurun_out/r3/tr_slots.log
