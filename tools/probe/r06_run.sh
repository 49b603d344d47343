python -c "import __graft_entry__ as G; G.smoke()" 2>&1 | grep -v amdgpu | tail -4
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06f_pytest_gpu.log 2>&1; grep -a "passed\|failed" gpurun_out/r06f_pytest_gpu.log | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/r06f_bench_20.log 2>gpurun_out/r06f_bench_20.err; tail -1 gpurun_out/r06f_bench_20.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value'], d['ms_per_step'], c['steady_step_ms'], c['render_step_ms'], d['roofline']['frac'], d['roofline'].get('traffic_source')); t=d['train_step']; print({k:t[k] for k in ('ms','library_graph_ms','cd360_ms','library_ms')}); print(d['cpu_baseline']['value'])"
