mkdir -p gpurun_out
for sg in 1 0; do
CAF_FUSED_STAGGER=$sg python bench.py --steps 5 --warmup 2 --no-cpu-baseline --engine fused > gpurun_out/bench_sg_$sg.json 2> gpurun_out/bench_sg_$sg.err || { tail -30 gpurun_out/bench_sg_$sg.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/bench_sg_$sg.json')); print('STAGGER=$sg', round(d['value'],1), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stages_ms_per_step'].items()})"
done
