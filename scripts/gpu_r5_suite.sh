#!/bin/bash
# Round 5: full GPU suite, smoke(), the default bench line
set -o pipefail
O=gpurun_out/r05_suite
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tee $O/gpu_suite.log | tail -5 || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > $O/bench_c2.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/r05_suite/bench_c2.json"))
print(j["engine"], round(j["value"],1), j["unit"], round(j["ms_per_step"],2), "ms/step")
print("roofline", {k:(round(v,3) if isinstance(v,float) else v) for k,v in j["roofline"].items() if k not in ("note","kernel")})
print("no_surface", round(j["no_surface"]["ms_per_step"],2), "surface_t", round(j["surface_t"]["ms_per_step"],2))
PY
