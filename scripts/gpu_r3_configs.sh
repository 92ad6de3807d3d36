#!/bin/bash
# round 3: timings of the BASELINE configurations for DESIGN.md (logs -> gpurun_out/r3/*.log)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3
timeout -k 10 300 python scripts/time_configs.py persistent tcc > gpurun_out/r3/configs_c3_c4share.log 2>&1
timeout -k 10 200 python scripts/time_tcc_literal.py > gpurun_out/r3/tcc_literal.log 2>&1
timeout -k 10 200 python scripts/time_c5.py > gpurun_out/r3/c5_zoom.log 2>&1
timeout -k 10 200 python scripts/time_long_template.py > gpurun_out/r3/long_template.log 2>&1
timeout -k 10 200 python scripts/time_perdelay.py > gpurun_out/r3/perdelay.log 2>&1
timeout -k 10 200 python scripts/time_fir.py > gpurun_out/r3/fir.log 2>&1
timeout -k 10 300 python scripts/bench_kernels.py > gpurun_out/r3/bench_kernels.log 2>&1
timeout -k 10 300 python bench.py --workload c4 --templates 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3/bench_c4_share.json 2> gpurun_out/r3/bench_c4_share.err
tail -n 40 gpurun_out/r3/configs_c3_c4share.log gpurun_out/r3/tcc_literal.log gpurun_out/r3/c5_zoom.log gpurun_out/r3/long_template.log gpurun_out/r3/perdelay.log gpurun_out/r3/fir.log gpurun_out/r3/bench_kernels.log
python -c "
import json; j=json.load(open('gpurun_out/r3/bench_c4_share.json')); print('c4 share', j['ms_per_step'], 'ms/step', j['value'], j['unit'])"
