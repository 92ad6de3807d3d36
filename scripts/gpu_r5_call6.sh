#!/bin/bash
# plan sweep of the run-time-compiled per-delay kernel: six processes, timing sections serialised by a lock
set -o pipefail
O=gpurun_out/r05_call8
mkdir -p $O
export CAF_JIT_CACHE=off SWEEP_LIMIT=40
python scripts/sweep_pdj_plans.py $O/sweep_a.csv 1200 5000 96 2400 > $O/a.log 2>&1 &
python scripts/sweep_pdj_plans.py $O/sweep_b.csv 1400 8000 360 1000 > $O/b.log 2>&1 &
python scripts/sweep_pdj_plans.py $O/sweep_c.csv 1536 12000 4096 100 > $O/c.log 2>&1 &
python scripts/sweep_pdj_plans.py $O/sweep_d.csv 1920 3600 1024 6000 > $O/d.log 2>&1 &
python scripts/sweep_pdj_plans.py $O/sweep_e.csv 3000 10000 256 7000 > $O/e.log 2>&1 &
python scripts/sweep_pdj_plans.py $O/sweep_f.csv 16384 2048 8192 640 > $O/f.log 2>&1 &
while [ -n "$(jobs -r)" ]; do sleep 30; echo "progress: $(cat $O/sweep_*.csv 2>/dev/null | wc -l) plans timed"; done
wait
cat $O/sweep_*.csv | wc -l
tail -3 $O/*.log
