#!/bin/bash
# tile role: hand-placed waits (CAF_PERSIST_MANUAL_WAIT=1) against compiler-placed ones (=0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/s12.log
for m in 1 0; do
  CAF_PERSIST_MANUAL_WAIT=$m timeout -k 10 120 python scripts/dbg_persist2.py > gpurun_out/d12_$m.log 2>&1 || { tail -30 gpurun_out/d12_$m.log; exit 1; }
  echo "manual=$m identical (of 6): $(grep -c 'differing: 0 \[\] tiles  | row results differing: 0' gpurun_out/d12_$m.log)" >> gpurun_out/s12.log
  for ts in ${TSS:-8 12}; do
    echo "== manual=$m tr_slots=$ts" >> gpurun_out/s12.log
    CAF_PERSIST_MANUAL_WAIT=$m CAF_PERSIST_DEBUG=1 CAF_PERSIST_TR_SLOTS=$ts timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --engine persistent 2>&1 | grep "workgroups (" | tail -2 >> gpurun_out/s12.log
    CAF_PERSIST_MANUAL_WAIT=$m CAF_PERSIST_TR_SLOTS=$ts timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --engine persistent 2>&1 | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   ', j['engine'], round(j['value'],1), 'Msamples/s', round(j['ms_per_step'],2), 'ms; kernel', round(j['stages_ms_per_step']['spectral_conj_multiply'],2))" >> gpurun_out/s12.log
  done
done
cat gpurun_out/s12.log
