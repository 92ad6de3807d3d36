#!/bin/bash
# second plan sweep (kernel with DPP group maxima): the first sweep's lengths + 24 more
set -o pipefail
O=gpurun_out/r05_call11
mkdir -p $O
export CAF_JIT_CACHE=off SWEEP_LIMIT=32
python scripts/sweep_pdj_plans.py $O/sweep_a.csv 1200 5000 96 2400 1500 4000 128 600 > $O/a.log 2>&1 &
python scripts/sweep_pdj_plans.py $O/sweep_b.csv 1400 8000 360 1000 1800 4800 192 720 > $O/b.log 2>&1 &
python scripts/sweep_pdj_plans.py $O/sweep_c.csv 1536 12000 4096 100 2000 6400 320 768 > $O/c.log 2>&1 &
python scripts/sweep_pdj_plans.py $O/sweep_d.csv 1920 3600 1024 6000 2500 9600 480 800 > $O/d.log 2>&1 &
python scripts/sweep_pdj_plans.py $O/sweep_e.csv 3000 10000 256 7000 3200 14336 500 900 > $O/e.log 2>&1 &
python scripts/sweep_pdj_plans.py $O/sweep_f.csv 16384 2048 8192 640 15000 16200 512 960 > $O/f.log 2>&1 &
while [ -n "$(jobs -r)" ]; do sleep 30; echo "progress: $(cat $O/sweep_*.csv 2>/dev/null | wc -l) plans timed"; done
wait
cat $O/sweep_*.csv | wc -l
tail -2 $O/*.log
