#!/bin/bash
# Round 5, W3: one pass over the variants of scripts/diag_stall.py (one process each), then the same call under the
# runtime's own log and under malloc settings that keep host memory mapped.
set -o pipefail
O=gpurun_out/r05_stall
mkdir -p $O
for v in base steps keep devin noget reuse small; do
  echo "== $v" | tee -a $O/variants.log
  timeout -k 10 120 python scripts/diag_stall.py $v 8 2>&1 | tee -a $O/variants.log
done
echo "== base, MALLOC_MMAP_THRESHOLD_=1 GiB, MALLOC_TRIM_THRESHOLD_=4 GiB, MALLOC_TOP_PAD_=256 MiB (host arrays come from the heap, the heap never shrinks)" | tee -a $O/variants.log
MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=4294967296 MALLOC_TOP_PAD_=268435456 timeout -k 10 120 python scripts/diag_stall.py base 8 2>&1 | tee -a $O/variants.log
echo "== base, GPU_PINNED_MIN_XFER_SIZE raised (copies staged, user pages not pinned)" | tee -a $O/variants.log
GPU_PINNED_MIN_XFER_SIZE=1073741824 timeout -k 10 120 python scripts/diag_stall.py base 8 2>&1 | tee -a $O/variants.log
echo "== base, HIP_HOST_COHERENT / HSA_USERPTR_FOR_PAGED_MEM=0" | tee -a $O/variants.log
HSA_USERPTR_FOR_PAGED_MEM=0 timeout -k 10 120 python scripts/diag_stall.py base 8 2>&1 | tee -a $O/variants.log
echo "== steps under AMD_LOG_LEVEL=4 (5 calls; last 24 MB of the log kept)" | tee -a $O/variants.log
AMD_LOG_LEVEL=4 timeout -k 10 300 python scripts/diag_stall.py steps 5 > $O/amdlog_steps.out 2> $O/amdlog_steps.err.full
tail -c 24000000 $O/amdlog_steps.err.full > $O/amdlog_steps.err; rm -f $O/amdlog_steps.err.full
cat $O/amdlog_steps.out | tee -a $O/variants.log
ls -la $O
