#!/bin/bash
# Round 5: staging transfers + host surface routing: new tests, the GPU suite, the stall variants again, transfer rates
set -o pipefail
O=gpurun_out/r05_call2
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_host_transfers.py -x -q 2>&1 | tee $O/test_host_transfers.log || exit 1
for v in base steps; do
  echo "== $v" | tee -a $O/stall_after.log
  timeout -k 10 120 python scripts/diag_stall.py $v 8 2>&1 | tee -a $O/stall_after.log
done
timeout -k 10 300 python scripts/time_cztxcorr.py rule rows engine 2>&1 | tee $O/timing_cztxcorr.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tee $O/gpu_suite.log
