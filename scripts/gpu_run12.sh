#!/bin/bash
# tile-role tuning: stagger of the odd waves (role statistics + end-to-end)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 120 python scripts/dbg_persist2.py > gpurun_out/d12.log 2>&1 || { tail -30 gpurun_out/d12.log; exit 1; }
echo "identical (of 6): $(grep -c 'differing: 0 \[\] tiles  | row results differing: 0' gpurun_out/d12.log)"
: > gpurun_out/s12.log
for g in ${STAGGERS:-0 2 4 8}; do
  echo "== stagger=$g tr_slots=${TS:-12}" >> gpurun_out/s12.log
  CAF_PERSIST_STAGGER=$g CAF_PERSIST_DEBUG=1 CAF_PERSIST_TR_SLOTS=${TS:-12} timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --engine persistent 2>&1 | grep "workgroups (" | tail -2 >> gpurun_out/s12.log
  CAF_PERSIST_STAGGER=$g CAF_PERSIST_TR_SLOTS=${TS:-12} timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --engine persistent 2>&1 | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   ', j['engine'], round(j['value'],1), 'Msamples/s', round(j['ms_per_step'],2), 'ms; kernel', round(j['stages_ms_per_step']['spectral_conj_multiply'],2))" >> gpurun_out/s12.log
done
cat gpurun_out/s12.log
