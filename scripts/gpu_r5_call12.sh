#!/bin/bash
set -o pipefail
O=gpurun_out/r05_call12
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_zoom_ingest.py tests/test_gpu_firos.py tests/test_gpu_kernels_fuzz.py -x -q 2>&1 | tail -4 || exit 1
for nt in 256 128; do
  echo "== CAF_FIR_POLY_NT=$nt" | tee -a $O/ab_fir_poly_nt.log
  CAF_FIR_POLY_NT=$nt timeout -k 10 120 python scripts/time_iq16_frontend.py 2>&1 | grep "front end" | tee -a $O/ab_fir_poly_nt.log
done
timeout -k 10 300 python scripts/time_fir.py 2>&1 | grep -i "upfirdn" | tee $O/timing_upfirdn.log
