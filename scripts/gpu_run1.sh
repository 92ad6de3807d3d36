mkdir -p gpurun_out
set -x
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -30 gpurun_out/bench_default.err; exit 1; }
cat gpurun_out/bench_default.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stages_ms_per_step'], d['roofline'], d.get('cpu_baseline'))"
for lb in 14 15 16 17; do for nb in 1 2 4; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --log2-block $lb --blocks-per-batch $nb 2>>gpurun_out/sweep.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lb',$lb,'nb',$nb, round(d['value'],1), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stages_ms_per_step'].items()})" >> gpurun_out/sweep.log
done; done
cat gpurun_out/sweep.log
