mkdir -p gpurun_out
for trf in 64 128; do
CAF_TR_F=$trf python bench.py --steps 5 --warmup 2 --no-cpu-baseline --engine fused > gpurun_out/bench_trf_$trf.json 2> gpurun_out/bench_trf_$trf.err || { tail -30 gpurun_out/bench_trf_$trf.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/bench_trf_$trf.json')); print('TR_F=$trf', round(d['value'],1), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stages_ms_per_step'].items()}, round(d['roofline_hbm_kernel']['frac'],3))"
done
