#!/bin/bash
# persistent engine: parity tests, then a sweep of the number of tile-first workgroups per XCD
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "engines_agree" > gpurun_out/t9.log 2>&1 || { tail -30 gpurun_out/t9.log; exit 1; }
tail -3 gpurun_out/t9.log
: > gpurun_out/b9.log
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --engine fused >> gpurun_out/b9.log 2>&1
for s in 0 4 6 8 10 12; do
  echo "tr_slots=$s" >> gpurun_out/b9.log
  CAF_PERSIST_TR_SLOTS=$s timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --engine persistent >> gpurun_out/b9.log 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/b9.log"):
    l=l.strip()
    if l.startswith("tr_slots"): print(l); continue
    if l.startswith("{"):
        j=json.loads(l); print(j["engine"], round(j["value"],1), "Msamples/s", round(j["ms_per_step"],2), "ms", {k:round(v,3) for k,v in j["stages_ms_per_step"].items()})
PY
