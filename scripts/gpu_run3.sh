mkdir -p gpurun_out
for eng in fused rocfft; do
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --engine $eng > gpurun_out/bench_$eng.json 2> gpurun_out/bench_$eng.err || { tail -30 gpurun_out/bench_$eng.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/bench_$eng.json')); print('$eng', round(d['value'],1), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stages_ms_per_step'].items()}, d['roofline'], d['roofline_hbm_kernel']['frac'])"
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --engine fused --no-surface > gpurun_out/bench_fused_nosurf.json 2>> gpurun_out/bench_fused.err
python -c "import json; d=json.load(open('gpurun_out/bench_fused_nosurf.json')); print('fused-nosurf', round(d['value'],1), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stages_ms_per_step'].items()})"
