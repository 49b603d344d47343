b() { python bench.py --steps 20 --warmup 5 --no-train-step --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['config']['steady_step_ms'], d['config']['render_step_ms'])"; }
for i in 1 2 3; do
b new
CD360_LIB=$PWD/custom-diffusion360_amd/lib/libcd360_oldgn.so b oldgn
done
python -m pytest tests/test_kernels_gpu.py -x -q -k "gn or group" 2>&1 | tail -2
