#!/bin/bash
# round-3 end: full GPU suite, smoke(), default bench (with the CPU baselines) -> gpurun_out/r3/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3/final_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r3/final_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/r3/bench_c2_final.json 2> gpurun_out/r3/bench_c2_final.err || { tail -5 gpurun_out/r3/bench_c2_final.err; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/r3/bench_c2_final.json"))
print(j["engine"], round(j["value"],1), j["unit"], round(j["ms_per_step"],2), "ms/step; no surface", round(j["no_surface"]["ms_per_step"],2))
print("roofline", {k:(round(v,3) if isinstance(v,float) else v) for k,v in j["roofline"].items() if k not in ("note","kernel")})
print("cpu_baseline", round(j["cpu_baseline"]["value"],4), j["cpu_baseline"]["cores"], "| threaded", round(j["cpu_baseline_threaded"]["value"],4), j["cpu_baseline_threaded"]["cores"], "| same algorithm", round(j["cpu_baseline_same_algorithm"]["value"],4), j["cpu_baseline_same_algorithm"]["cores"])
PY
