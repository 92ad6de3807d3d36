#!/bin/bash
# full GPU suite + tr_slots sweep of the persistent engine
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/t13.log 2>&1
rc=$?
tail -5 gpurun_out/t13.log
[ $rc -ne 0 ] && exit $rc
: > gpurun_out/b13.log
for s in ${SLOTS:-8 10 12 14 16 20}; do
  echo "tr_slots=$s" >> gpurun_out/b13.log
  CAF_PERSIST_TR_SLOTS=$s timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --engine persistent 2>&1 | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   ', j['engine'], round(j['value'],1), 'Msamples/s', round(j['ms_per_step'],2), 'ms; kernel', round(j['stages_ms_per_step']['spectral_conj_multiply'],2))" >> gpurun_out/b13.log
done
cat gpurun_out/b13.log
