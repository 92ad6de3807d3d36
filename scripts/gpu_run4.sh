mkdir -p gpurun_out
for th in 1024 512; do
CAF_FUSED_THREADS=$th python bench.py --steps 5 --warmup 2 --no-cpu-baseline --engine fused > gpurun_out/bench_fused_$th.json 2> gpurun_out/bench_fused_$th.err || { tail -30 gpurun_out/bench_fused_$th.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/bench_fused_$th.json')); print('fused threads=$th', round(d['value'],1), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stages_ms_per_step'].items()}, round(d['roofline']['frac'],3), round(d['roofline_hbm_kernel']['frac'],3))"
done
