#!/bin/bash
# per-CU streaming rate of the tile role alone at different concurrency
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
CAF_PERSIST_TILE_ONLY="8,16,32,64,96,128,192,256" CAF_PERSIST_DEBUG=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --engine persistent 2>&1 | grep "tile role alone\|workgroups (" > gpurun_out/s15.log
cat gpurun_out/s15.log
