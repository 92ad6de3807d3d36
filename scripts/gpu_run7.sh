mkdir -p gpurun_out
for nt in 1 0; do
CAF_NT_STORE=$nt python bench.py --steps 5 --warmup 2 --no-cpu-baseline --engine fused > gpurun_out/bench_nt_$nt.json 2> gpurun_out/bench_nt_$nt.err || { tail -30 gpurun_out/bench_nt_$nt.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/bench_nt_$nt.json')); print('NT=$nt', round(d['value'],1), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stages_ms_per_step'].items()}, round(d['roofline_hbm_kernel']['frac'],3))"
done
