mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/t2.log 2>&1 || { tail -30 gpurun_out/t2.log; exit 1; }
tail -2 gpurun_out/t2.log
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err || { tail -30 gpurun_out/bench_r01.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/bench_r01.json')); print(d['value'], d['ms_per_step'], d['stages_ms_per_step'], d['roofline'], d.get('cpu_baseline'))"
bash scripts/gpu_profile.sh r01
