#!/bin/bash
# persistent engine: residency robustness (more / fewer workgroups than CUs), then the rocprofv3 trace + PMC passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/r14.log
for w in 64 512 1000; do
  echo "== CAF_PERSIST_WGS=$w" >> gpurun_out/r14.log
  CAF_PERSIST_WGS=$w timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   ', j['engine'], round(j['value'],1), 'Msamples/s', round(j['ms_per_step'],2), 'ms')" >> gpurun_out/r14.log 2>&1
  echo "   rc=$?" >> gpurun_out/r14.log
done
cat gpurun_out/r14.log
bash scripts/gpu_profile.sh r01_persistent && timeout -k 10 300 python bench.py > gpurun_out/bench_persistent.json 2> gpurun_out/bench_persistent.err; tail -c 1500 gpurun_out/bench_persistent.json
