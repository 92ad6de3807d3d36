#!/bin/bash
# round-end rehearsal: full GPU suite, smoke(), default bench (with the CPU baselines), rocprofv3 trace + PMC passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1
rc=$?
tail -4 gpurun_out/final_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || { tail -5 gpurun_out/bench_final.err; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/bench_final.json"))
print(j["engine"], round(j["value"],1), j["unit"], round(j["ms_per_step"],2), "ms/step")
print("roofline", {k:(round(v,3) if isinstance(v,float) else v) for k,v in j["roofline"].items() if k not in ("note","kernel")})
print("cpu_baseline", round(j["cpu_baseline"]["value"],4), j["cpu_baseline"]["cores"], "| same algorithm", round(j["cpu_baseline_same_algorithm"]["value"],4), j["cpu_baseline_same_algorithm"]["cores"])
PY
bash scripts/gpu_profile.sh r01_final2 > gpurun_out/profile_final.log 2>&1 || { tail -20 gpurun_out/profile_final.log; exit 1; }
grep -n "k_caf_persistent" gpurun_out/prof_r01_final2/summary.txt | cut -c1-150
