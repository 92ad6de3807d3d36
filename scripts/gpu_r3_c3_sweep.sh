#!/bin/bash
# round 3: C3 (64 templates x 1 bin, rows + peak) against the hypotheses per FFT work item
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
LOG=gpurun_out/r3/c3_hyp_per_wg_sweep.log
: > $LOG
for h in 0 8 16 32 64 0; do
  echo "== CAF_HYP_PER_WG=$h (0: the plan's own choice)" >> $LOG
  if [ $h = 0 ]; then unset CAF_HYP_PER_WG; else export CAF_HYP_PER_WG=$h; fi
  timeout -k 10 200 python scripts/time_configs.py persistent 2>&1 | grep "^C3\|^C4" >> $LOG || exit 1
done
cat $LOG
