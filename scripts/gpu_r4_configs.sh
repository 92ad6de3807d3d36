#!/bin/bash
# round 4: timings of the BASELINE configurations and of the reference's benchmark shapes at head (logs -> gpurun_out/r4/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4
timeout -k 10 300 python scripts/time_configs.py persistent tcc > gpurun_out/r4/timing_configs_c3_c4share.log 2>&1
timeout -k 10 200 python scripts/time_tcc_literal.py > gpurun_out/r4/timing_tcc_literal.log 2>&1
timeout -k 10 200 python scripts/time_c5.py > gpurun_out/r4/timing_c5_zoom.log 2>&1
timeout -k 10 300 python scripts/time_perdelay.py > gpurun_out/r4/timing_perdelay.log 2>&1
timeout -k 10 200 python scripts/time_fir.py > gpurun_out/r4/timing_fir.log 2>&1
timeout -k 10 300 python scripts/bench_kernels.py > gpurun_out/r4/timing_bench_kernels.log 2>&1
timeout -k 10 200 python scripts/time_small_calls.py > gpurun_out/r4/timing_small_calls.log 2>&1
timeout -k 10 200 python scripts/time_surface_t.py > gpurun_out/r4/timing_surface_t.log 2>&1
timeout -k 10 400 python scripts/time_long_template.py > gpurun_out/r4/timing_long_template.log 2>&1
timeout -k 10 300 python bench.py --workload c4 --templates 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4/bench_c4_one_share_64tmpl.json 2> gpurun_out/r4/bench_c4_share.err
CAF_PERSIST_DEBUG=2 timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-side-figure > /dev/null 2> gpurun_out/r4/persistent_role_split_c2.log
grep -v amdgpu gpurun_out/r4/timing_configs_c3_c4share.log gpurun_out/r4/timing_tcc_literal.log gpurun_out/r4/timing_c5_zoom.log gpurun_out/r4/timing_fir.log gpurun_out/r4/timing_small_calls.log | tail -n 40
python -c "
import json; j=json.load(open('gpurun_out/r4/bench_c4_one_share_64tmpl.json')); print('c4 share', j['ms_per_step'], 'ms/step', j['value'], j['unit'])"
tail -4 gpurun_out/r4/persistent_role_split_c2.log
