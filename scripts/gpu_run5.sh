mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/t6.log 2>&1; echo "rc=$?" >> gpurun_out/t6.log; tail -4 gpurun_out/t6.log
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r01b.json 2> gpurun_out/bench_r01b.err || { tail -30 gpurun_out/bench_r01b.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/bench_r01b.json')); print(d['engine'], d['value'], d['ms_per_step'], d['stages_ms_per_step'], d['roofline'], d['roofline_hbm_kernel'], d.get('cpu_baseline'))"
bash scripts/gpu_profile.sh r01_fused
