python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --no-train-step --no-cpu-baseline > gpurun_out/r06d_bench_20.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r06d_bench_20.json')); c=d['config']; print(d['value'], d['ms_per_step'], c['steady_step_ms'], c['render_step_ms']); print({k:(v['frac'],v['avg_us']) for k,v in d['rooflines'].items() if 'avg_us' in v})"
