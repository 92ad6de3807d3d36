#!/bin/bash
# role statistics + end-to-end timing of the persistent engine on C2 for several splits (after a bitwise
# check against the two-launch engine)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 120 python scripts/dbg_persist2.py > gpurun_out/d11.log 2>&1 || { tail -30 gpurun_out/d11.log; exit 1; }
echo "identical (of 6): $(grep -c 'differing: 0 \[\] tiles  | row results differing: 0' gpurun_out/d11.log)"
: > gpurun_out/s11.log
for s in ${SLOTS:-0 7 12}; do
  echo "== tr_slots=$s" >> gpurun_out/s11.log
  CAF_PERSIST_DEBUG=1 CAF_PERSIST_TR_SLOTS=$s timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --engine persistent 2>&1 | grep "workgroups (" | tail -2 >> gpurun_out/s11.log
  CAF_PERSIST_TR_SLOTS=$s timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --engine persistent 2>&1 | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   ', j['engine'], round(j['value'],1), 'Msamples/s', round(j['ms_per_step'],2), 'ms; kernel', round(j['stages_ms_per_step']['spectral_conj_multiply'],2))" >> gpurun_out/s11.log
done
cat gpurun_out/s11.log
