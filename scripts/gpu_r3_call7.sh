#!/bin/bash
# round 3, call 7: are the |y|^2 tiles served from the Infinity Cache when they are not marked non-temporal?  items of 64 / 32 / 16
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
LOG=gpurun_out/r3/tile_policy.log
: > $LOG
for hp in 64 32 16; do
for lib in libcaf libcaf_tnt0 libcaf libcaf_tnt0; do
  echo "== $lib CAF_HYP_PER_WG=$hp" >> $LOG
  env CAF_LIBRARY=$lib CAF_HYP_PER_WG=$hp timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side-figure 2>&1 | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('   ', round(j['ms_per_step'],2), 'ms; kernel', round(j['stages_ms_per_step']['spectral_conj_multiply'],3))" >> $LOG || exit 1
done
done
cat $LOG
