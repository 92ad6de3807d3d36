#!/bin/bash
# round 5: timings of the BASELINE configurations and of the reference's benchmark shapes at head (logs -> gpurun_out/r5/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
export CAF_JIT_CACHE=off
timeout -k 10 300 python scripts/time_configs.py persistent tcc > gpurun_out/r5/timing_configs_c3_c4share.log 2>&1
timeout -k 10 200 python scripts/time_tcc_literal.py > gpurun_out/r5/timing_tcc_literal.log 2>&1
timeout -k 10 200 python scripts/time_c5.py > gpurun_out/r5/timing_c5_zoom.log 2>&1
timeout -k 10 300 python scripts/time_perdelay.py > gpurun_out/r5/timing_perdelay.log 2>&1
timeout -k 10 300 python scripts/time_perdelay_mixed.py 1200 1400 5000 1536 1920 3000 3600 8000 12000 96 360 1430 2431 10000 16384 18000 20000 > gpurun_out/r5/timing_perdelay_jit.log 2>&1
timeout -k 10 200 python scripts/time_fir.py > gpurun_out/r5/timing_fir.log 2>&1
timeout -k 10 300 python scripts/bench_kernels.py > gpurun_out/r5/timing_bench_kernels.log 2>&1
timeout -k 10 200 python scripts/time_small_calls.py > gpurun_out/r5/timing_small_calls.log 2>&1
timeout -k 10 200 python scripts/time_cztxcorr.py rule rows engine > gpurun_out/r5/timing_cztxcorr.log 2>&1
timeout -k 10 400 python scripts/time_long_template.py > gpurun_out/r5/timing_long_template.log 2>&1
timeout -k 10 300 python scripts/time_reference_benchmarks.py > gpurun_out/r5/timing_reference_benchmarks_full_size.log 2>&1
timeout -k 10 300 python bench.py --workload c4 --templates 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5/bench_c4_one_share_64tmpl.json 2> gpurun_out/r5/bench_c4_share.err
BENCH_REHEARSE_GLOO=1 timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --rx-log2 22 > gpurun_out/r5/bench_two_rank_rehearsal.json 2> gpurun_out/r5/bench_two_rank.err
BENCH_REHEARSE_GLOO=1 timeout -k 10 300 python bench.py --gpus 2 --shard freq --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5/bench_freq_shard_rehearsal.json 2> gpurun_out/r5/bench_freq_shard.err
CAF_PERSIST_DEBUG=2 timeout -k 10 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-side-figure > /dev/null 2> gpurun_out/r5/persistent_role_split_c2.log
grep -v amdgpu gpurun_out/r5/timing_configs_c3_c4share.log gpurun_out/r5/timing_tcc_literal.log gpurun_out/r5/timing_c5_zoom.log gpurun_out/r5/timing_cztxcorr.log | tail -n 30
python -c "
import json
j=json.load(open('gpurun_out/r5/bench_c4_one_share_64tmpl.json')); print('c4 share', j['ms_per_step'], 'ms/step', j['value'], j['unit'])
j=json.load(open('gpurun_out/r5/bench_two_rank_rehearsal.json')); print('two ranks (gloo rehearsal)', j['ms_per_step'], j.get('per_rank_ms_per_step'), j.get('peak_table_allgather_ms'))
j=json.load(open('gpurun_out/r5/bench_freq_shard_rehearsal.json')); print('freq shard (gloo rehearsal)', j['ms_per_step'], j.get('per_rank_ms_per_step'))"
tail -4 gpurun_out/r5/persistent_role_split_c2.log
